// Explicit instance of nlr_mlp_kernel<NLR_INST_WT, 8, 2, NLR_INST_HT, NLR_INST_PREC> (see nlr_mlp_kernel.h).
#include "nlr_mlp_kernel.h"

NLR_MLP_DECLARE(NLR_INST_WT, NLR_INST_HT, NLR_INST_PREC, NLR_INST_COMP) {
    hipLaunchKernelGGL((nlr_mlp_kernel<NLR_INST_WT, 8, 2, NLR_INST_HT, NLR_INST_PREC, (NLR_INST_COMP != 0)>), grid, dim3(256), 0, st, P);
}

#ifdef NLR_STAMPS
// diagnostic builds only (scripts/diag_build.sh builds ONE instance with -DNLR_STAMPS): phase stamps of nlr_mlp_kernel, read back by
// scripts/stamp_probe.py.  Device symbols do not link across translation units, so buffer and reader live beside the instance.
__device__ unsigned long long nlr_stamp_buf[1024 * NLR_NSTAMP];
extern "C" int nlr_debug_stamps(unsigned long long *host_out, size_t count) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nlr_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
