// Explicit instance of nlr_mlp_kernel<NLR_INST_WT, 8, 2, NLR_INST_HT, NLR_INST_PREC> (see nlr_mlp_kernel.h).
#include "nlr_mlp_kernel.h"

NLR_MLP_DECLARE(NLR_INST_WT, NLR_INST_HT, NLR_INST_PREC) {
    hipLaunchKernelGGL((nlr_mlp_kernel<NLR_INST_WT, 8, 2, NLR_INST_HT, NLR_INST_PREC>), grid, dim3(256), 0, st, P);
}
