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
    if (i == 31 && P.dnorm) {  // |directions| (render.py:176: delta = t_delta * ||d||), same expression as nlr_composite_kernel
        const float dx = P.dirs[(size_t)ray * 3], dy = P.dirs[(size_t)ray * 3 + 1], dz = P.dirs[(size_t)ray * 3 + 2];
        P.dnorm[ray] = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
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
// silently emulated.  X(view width / 32, head units of 32, precision, compositing mode)
#define NLR_FOR_ALL_INSTANCES(X) \
    X(8, 4, 0, 0) X(8, 4, 1, 0) X(8, 4, 2, 0) X(8, 4, 2, 1) X(8, 2, 0, 0) X(8, 2, 1, 0) X(8, 2, 2, 0) X(8, 2, 2, 1) \
    X(8, 0, 0, 0) X(8, 0, 1, 0) X(8, 0, 2, 0) X(8, 0, 2, 1) X(4, 4, 0, 0) X(4, 4, 1, 0) X(4, 4, 2, 0) X(4, 4, 2, 1) \
    X(4, 2, 0, 0) X(4, 2, 1, 0) X(4, 2, 2, 0) X(4, 2, 2, 1) X(4, 0, 0, 0) X(4, 0, 1, 0) X(4, 0, 2, 0) X(4, 0, 2, 1)
#define NLR_DECL(wt, ht, pr, cm) NLR_MLP_DECLARE(wt, ht, pr, cm);
NLR_FOR_ALL_INSTANCES(NLR_DECL)
#undef NLR_DECL

static bool have_instance(uint32_t W, uint32_t HT, uint32_t prec, uint32_t comp) {
#define NLR_HAS(wt, ht, pr, cm) \
    if (W == wt * 32 && HT == ht && prec == pr && comp == cm) return true;
    NLR_FOR_ALL_INSTANCES(NLR_HAS)
#undef NLR_HAS
    return false;
}

// Compositing mode needs: an instance; whole 32-sample segments per ray that the compositing kernel's lane layout can address
// (its lane owns per = ceil(S / 64) consecutive samples: a segment must start on a lane); every record slot inside 32 floats
// ([0,K) classes, [K] intensity, [29,32) rgb); inputs that go through the LDS staging path.
bool nlr_mlp_can_composite(uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, uint32_t F, uint32_t S, uint32_t K, bool use_int, uint64_t M) {
    const uint32_t per = (S + 63) / 64;
    return WB == 256 && have_instance(W, HT, prec, 1) && S % 32 == 0 && per > 0 && 32 % per == 0 && K + (use_int ? 1u : 0u) <= 29 &&
           F % 4 == 0 && F >= 4 && F <= 4 * NLR_STAGE_PIECES && M >= 32;
}

int nlr_launch_mlp(const MlpParams &P, uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, uint32_t cus, hipStream_t st) {
    NLR_CHECK_ARG(P.M > 0, "mlp: no samples");
    NLR_CHECK_ARG(P.tape && P.tape_chunks > 0, "mlp: weight tape missing");
    NLR_CHECK_ARG(P.bias_all && P.bias_count <= 3072 && P.bias_count % 4 == 0, "mlp: bias block missing or > 3072 floats");
    NLR_CHECK_ARG(cus > 0, "mlp: CU count of the model's device is unknown");
    const uint32_t FT = (P.F + 31) / 32;
    const uint32_t comp = P.seg ? 1 : 0;
    if (comp)
        NLR_CHECK_ARG(P.tdist && P.dnorm && P.rgb && P.feat_piece_major && nlr_mlp_can_composite(W, WB, HT, prec, P.F, P.S, P.K, P.inten != nullptr, P.M),
                      "mlp: compositing mode is not available for this configuration");
    // persistent workgroups: one per CU (~156 KiB of LDS each, so one is all a CU holds), tiles of 256 samples round-robin
    const uint32_t ntiles = (P.M + NLR_TILE - 1) / NLR_TILE;
    dim3 grid(ntiles < cus ? ntiles : cus);
    if (WB == 256 && FT <= 2 && P.F % 4 == 0) {  // (instances are built for two k-blocks; a tape of F <= 32 features is padded to two)
#define NLR_TRY(wt, ht, pr, cm)                                   \
    if (W == wt * 32 && HT == ht && prec == pr && comp == cm) {   \
        NLR_MLP_LAUNCH_NAME(wt, ht, pr, cm)(P, grid, st);         \
        NLR_LAUNCH_CHECK("nlr_mlp_kernel");                       \
        return NLR_OK;                                            \
    }
        NLR_FOR_ALL_INSTANCES(NLR_TRY)
#undef NLR_TRY
    }
    NLR_FAIL(NLR_ERR_UNSUPPORTED,
             "NerfMLP shape (width %u, bottleneck %u, %u grid features, %u head tiles, precision %u) has no fused kernel "
             "instance; built: view widths 128 and 256 with 0, 1 or 2 heads (semantic / intensity), bottleneck 256, 4..64 grid features (multiple of 4)",
             W, WB, P.F, HT, prec);
}
