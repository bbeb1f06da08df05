// Host launcher of the NerfMLP kernel and the per-ray direction-encoding pre-kernel (kernel: nlr_mlp_kernel.h).
#include "nlr_mlp_kernel.h"

// ---------------------------------------------------------------------------------------------
// per-ray direction encoding pos_enc(viewdirs, 0, deg, append_identity) (coord.py:199-210), zero-padded to 32
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nlr_direnc_kernel(DirEncParams P) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.N * 32) return;
    const uint32_t ray = t >> 5;
    const int i = t & 31;
    float val = 0.0f;
    if (i < 3) {
        val = P.viewdirs[(size_t)ray * 3 + i];
    } else if (i < (int)P.E) {  // [x, sin(2^k x), sin(2^k x + pi/2)], k-major
        const int q = i - 3;
        const int half = q >= (int)(3 * P.deg);
        const int r = half ? q - 3 * P.deg : q;
        const int k = r / 3, c = r - 3 * k;
        const float sx = P.viewdirs[(size_t)ray * 3 + c] * (float)(1u << k);
        val = half ? sinf(sx + 1.57079637050628662f) : sinf(sx);
    }
    P.out[t] = val;
}

int nlr_launch_direnc(const DirEncParams &P, hipStream_t st) {
    NLR_CHECK_ARG(P.E <= 32, "direction encoding has %u > 32 features (deg_view > 4) -- no fused path", P.E);
    const uint32_t T = P.N * 32;
    hipLaunchKernelGGL(nlr_direnc_kernel, dim3((T + 255) / 256), dim3(256), 0, st, P);
    NLR_LAUNCH_CHECK("nlr_direnc_kernel");
    return NLR_OK;
}

// Supported shapes are instantiated explicitly (one translation unit each); everything else is reported, not
// silently emulated.
#define NLR_FOR_ALL_INSTANCES(X) X(8, 4, 0) X(8, 4, 1) X(8, 4, 2) X(8, 2, 0) X(8, 2, 1) X(8, 2, 2) X(4, 2, 0) X(4, 2, 1) X(4, 2, 2)
#define NLR_DECL(wt, ht, pr) NLR_MLP_DECLARE(wt, ht, pr);
NLR_FOR_ALL_INSTANCES(NLR_DECL)
#undef NLR_DECL

int nlr_launch_mlp(const MlpParams &P, uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, uint32_t cus, hipStream_t st) {
    NLR_CHECK_ARG(P.M > 0, "mlp: no samples");
    NLR_CHECK_ARG(P.tape && P.tape_chunks > 0, "mlp: weight tape missing");
    NLR_CHECK_ARG(P.bias_all && P.bias_count <= 4096 && P.bias_count % 4 == 0, "mlp: bias block missing or > 4096 floats");
    const uint32_t FT = (P.F + 31) / 32;
    // persistent workgroups: one per CU (112 KiB of LDS each, so one is all a CU holds), tiles of 256 samples round-robin
    NLR_CHECK_ARG(cus > 0, "mlp: CU count of the model's device is unknown");
    const uint32_t ntiles = (P.M + NLR_TILE - 1) / NLR_TILE;
    dim3 grid(ntiles < cus ? ntiles : cus);
    if (WB == 256 && FT == 2 && P.F % 4 == 0) {
#define NLR_TRY(wt, ht, pr)                                   \
    if (W == wt * 32 && HT == ht && prec == pr) {             \
        NLR_MLP_LAUNCH_NAME(wt, ht, pr)(P, grid, st);         \
        NLR_LAUNCH_CHECK("nlr_mlp_kernel");                   \
        return NLR_OK;                                        \
    }
        NLR_FOR_ALL_INSTANCES(NLR_TRY)
#undef NLR_TRY
    }
    NLR_FAIL(NLR_ERR_UNSUPPORTED,
             "NerfMLP shape (width %u, bottleneck %u, %u grid features, %u head tiles, precision %u) has no fused kernel "
             "instance; built: (width 256, sem+intensity), (width 256, sem), (width 128, sem), bottleneck 256, 33..64 grid features",
             W, WB, P.F, HT, prec);
}
