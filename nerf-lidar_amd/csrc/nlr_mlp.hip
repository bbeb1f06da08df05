// NerfMLP evaluation on the matrix cores: density trunk -> semantic/intensity heads -> view MLP -> rgb.
//
// Replaces (rows a-9..a-12 of the scope table):
//   ZI/models.py:887-889, 996-997, 1116     density_layer (F->64->256), softplus(raw - 1)
//   ZI/models.py:954-961, 1124-1143         sem_layer (256->64->19, softmax), intensity_layer (256->64->1)
//   ZI/coord.py:199-210, models.py:1190-1196  pos_enc(viewdirs) broadcast over samples
//   ZI/models.py:939-951, 1223-1234, 1251   lin_second_stage_i (+skip concat after layer 0), rgb_layer, sigmoid, padding
//
// Design (CDNA4, not a translation of the nn.Linear chain):
//   * the whole chain runs TRANSPOSED, activations^T = W . x^T, so that an MFMA result tile (32 output
//     features x 32 samples: sample on the lane, features in the 16 accumulator registers) is already
//     the B operand of the next layer's MFMA (cdna_hip_programming.md section 3, "An accumulator tile as
//     the next MFMA's operand").  Activations never leave the register file: no LDS round trip, no
//     barrier between the 8+ layers.  One wavefront owns 32 samples end to end.
//   * weights are the A operand, pre-packed at model-create time into exactly the per-lane fragment
//     order (including the permuted k order the accumulator layout implies), so every fragment fetch is
//     one fully coalesced 16-byte-per-lane load of 1 KiB per wavefront.
//   * the direction encoding is constant per ray, so its 27 input columns of layers 0 and 1 are folded
//     into a per-ray bias by a small pre-kernel and every GEMM on the matrix cores has K % 32 == 0.
//   * precision: layers whose error reaches depth / semantic argmax / intensity (density trunk, heads)
//     use the exact-f32 MFMA (v_mfma_f32_32x32x2_f32); the view MLP (rgb only, 92 % of the MACs) uses
//     bf16 MFMA (v_mfma_f32_32x32x16_bf16) with f32 accumulation.  NLR_PREC_F32 runs everything in f32.
#include "nlr_kernels.h"


// row of accumulator register r for lane half h inside a 32-row tile
__device__ __forceinline__ int nlr_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int OT>
__device__ __forceinline__ void nlr_acc_bias(f32x16 (&acc)[OT], const float *__restrict__ bias, int h) {
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + o * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][q * 4 + e] = v[e];
        }
}

// ---- weight tape: global -> registers -> LDS (double buffered), shared by the 4 waves of a workgroup ----------
// A chunk is 16 KiB = 16 fragments of 1 KiB.  While the waves run the MFMAs of chunk c out of LDS buffer c&1, the
// 256 threads have the 4 x 16 B global loads of chunk c+1 in flight; they land in the other buffer after the MFMAs,
// one __syncthreads() per chunk.  Every wave needs every fragment (each wave owns 32 samples and all output
// features), so staging through LDS cuts the L2 -> CU weight traffic 4x against per-wave global loads and puts the
// fragment reads on ds_read_b128 (64-cycle latency) instead of L2 (500+ cycles at one wave per SIMD).
#define NLR_CHUNK_SLOTS 1024  // uint4 slots per chunk
struct Tape {
    const uint4 *__restrict__ base;
    uint4 *lds;  // [2][NLR_CHUNK_SLOTS]
    uint4 nxt[4];
    int cur, total, tid;
    __device__ __forceinline__ void load(int c) {
        const uint4 *p = base + (size_t)c * NLR_CHUNK_SLOTS + tid;
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[r] = p[r * 256];
    }
    __device__ __forceinline__ void store(int c) {
        uint4 *q = lds + (c & 1) * NLR_CHUNK_SLOTS + tid;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r * 256] = nxt[r];
    }
    __device__ __forceinline__ void prologue() {
        cur = 0;
        load(0);
        store(0);
        __syncthreads();
    }
    __device__ __forceinline__ void begin() {
        if (cur + 1 < total) load(cur + 1);
    }
    __device__ __forceinline__ void end() {
        if (cur + 1 < total) store(cur + 1);
        __syncthreads();
        ++cur;
    }
    template <typename T>
    __device__ __forceinline__ T frag(int f, int lane) const {
        return *reinterpret_cast<const T *>(lds + (cur & 1) * NLR_CHUNK_SLOTS + f * 64 + lane);
    }
};

// acc[o] += W[o-tile, :] . in   on the exact-f32 MFMA.  KG = number of 8-feature k-groups (one fragment = 4 k-steps).
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_f32(f32x16 (&acc)[OT], const f32x16 (&in)[KT], Tape &tp, int lane) {
    static_assert(KG <= KT * 4, "k-groups exceed the input tiles");
    constexpr int NF = KG * OT, NCH = (NF + 15) / 16;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        tp.begin();
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int idx = ch * 16 + f;
            if (idx < NF) {
                const int g = idx / OT, o = idx % OT;
                const f32x4 a = tp.frag<f32x4>(f, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], in[g >> 2][(g & 3) * 4 + e], acc[o], 0, 0, 0);
            }
        }
        tp.end();
    }
}

// acc[o] += W[o-tile, :] . in   on the bf16 MFMA.  KG = number of 16-feature k-steps (= 2 per input tile).
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_bf16(f32x16 (&acc)[OT], const TileH (&in)[KT], Tape &tp, int lane) {
    static_assert(KG <= KT * 2, "k-steps exceed the input tiles");
    constexpr int NF = KG * OT, NCH = (NF + 15) / 16;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        tp.begin();
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int idx = ch * 16 + f;
            if (idx < NF) {
                const int g = idx / OT, o = idx % OT;
                const bf16x8 a = tp.frag<bf16x8>(f, lane);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[g >> 1].f[g & 1], acc[o], 0, 0, 0);
            }
        }
        tp.end();
    }
}

// Split-bf16 ("bf16x3"): W = Wh + Wl, x = xh + xl (each part bf16), W.x ~= Wh.xh + Wh.xl + Wl.xh with f32
// accumulation: 16 mantissa bits per operand (relative error ~2^-16) at 3/16 of the exact-f32 MFMA cost.
// Fragments come in (hi, lo) pairs.
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_x3(f32x16 (&acc)[OT], const TileH (&inh)[KT], const TileH (&inl)[KT], Tape &tp, int lane) {
    static_assert(KG <= KT * 2, "k-steps exceed the input tiles");
    constexpr int NP = KG * OT, NCH = (NP + 7) / 8;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        tp.begin();
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            const int idx = ch * 8 + pr;
            if (idx < NP) {
                const int g = idx / OT, o = idx % OT;
                const bf16x8 ah = tp.frag<bf16x8>(2 * pr, lane);
                const bf16x8 al = tp.frag<bf16x8>(2 * pr + 1, lane);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, inh[g >> 1].f[g & 1], acc[o], 0, 0, 0);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, inl[g >> 1].f[g & 1], acc[o], 0, 0, 0);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, inh[g >> 1].f[g & 1], acc[o], 0, 0, 0);
            }
        }
        tp.end();
    }
}

template <int T, bool RELU>
__device__ __forceinline__ void nlr_pack(TileH (&dst)[T], const f32x16 (&src)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = src[t][8 * s + j];
                dst[t].f[s][j] = (__bf16)(RELU ? fmaxf(v, 0.0f) : v);
            }
}

// hi = bf16(x), lo = bf16(x - hi)   (x - hi is exact in f32)
template <int T, bool RELU>
__device__ __forceinline__ void nlr_split(TileH (&hi)[T], TileH (&lo)[T], const f32x16 (&src)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = src[t][8 * s + j];
                if (RELU) v = fmaxf(v, 0.0f);
                const __bf16 h = (__bf16)v;
                hi[t].f[s][j] = h;
                lo[t].f[s][j] = (__bf16)(v - (float)h);
            }
}

template <int T>
__device__ __forceinline__ void nlr_relu(f32x16 (&x)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = fmaxf(x[t][r], 0.0f);
}

// WT = view width / 32, BT = bottleneck / 32, FG = ceil(F/8), HT = head hidden tiles (0, 2 or 4)
// PREC: NLR_PREC_F32 (all f32), NLR_PREC_MIXED (trunk+heads f32, view bf16), NLR_PREC_FAST (trunk+heads bf16x3, view bf16)
template <int WT, int BT, int FG, int HT, int PREC>
__global__ void __launch_bounds__(256, 1) nlr_mlp_kernel(MlpParams P) {
    __shared__ __align__(16) uint4 lds_tape[2 * NLR_CHUNK_SLOTS];
    constexpr bool VIEW_F32 = (PREC == NLR_PREC_F32);
    constexpr bool X3 = (PREC == NLR_PREC_FAST);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const uint32_t sample = (blockIdx.x * 4 + wave) * 32 + col;
    const bool valid = sample < P.M;
    const uint32_t sc = valid ? sample : P.M - 1;
    constexpr int FT = (FG + 3) / 4;

    Tape tp;
    tp.base = P.tape;
    tp.lds = lds_tape;
    tp.total = (int)P.tape_chunks;
    tp.tid = threadIdx.x;
    tp.prologue();

    // ---- features -> accumulator-layout tiles (lane half h holds features 8q+4h..+3 of each group)
    f32x16 fin[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[t][r] = 0.0f;
    {
        const float *fp = P.feat + (size_t)sc * P.F;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            const uint32_t f0 = 8 * g + 4 * h;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (f0 + 4 <= P.F) v = *reinterpret_cast<const f32x4 *>(fp + f0);
#pragma unroll
            for (int e = 0; e < 4; ++e) fin[g >> 2][(g & 3) * 4 + e] = v[e];
        }
    }
    f32x16 hb[BT];
    TileH hbh[VIEW_F32 ? 1 : BT];  // bf16 copy of the bottleneck for the view MLP (= hi part in FAST mode)
    f32x16 lo[1];
    if constexpr (!X3) {
        // ---- density_layer.0 : F -> 64, ReLU
        f32x16 hid[2];
        nlr_acc_bias<2>(hid, P.b_d0, h);
        nlr_gemm_f32<2, FG, FT>(hid, fin, tp, lane);
        nlr_relu<2>(hid);
        // ---- density_layer.2 : 64 -> bottleneck (no activation); row 0 is the raw density
        nlr_acc_bias<BT>(hb, P.b_d2, h);
        nlr_gemm_f32<BT, 8, 2>(hb, hid, tp, lane);
        if constexpr (HT > 0) {
            f32x16 hh[HT];
            nlr_acc_bias<HT>(hh, P.b_h1, h);
            nlr_gemm_f32<HT, BT * 4, BT>(hh, hb, tp, lane);
            nlr_relu<HT>(hh);
            nlr_acc_bias<1>(lo, P.b_h2, h);
            nlr_gemm_f32<1, HT * 4, HT>(lo, hh, tp, lane);
        }
        if constexpr (!VIEW_F32) nlr_pack<BT, false>(hbh, hb);
    } else {
        constexpr int FK = (FG + 1) / 2;  // 16-feature k-steps covering the grid features
        TileH fh[FT], fl[FT];
        nlr_split<FT, false>(fh, fl, fin);
        f32x16 hid[2];
        nlr_acc_bias<2>(hid, P.b_d0, h);
        nlr_gemm_x3<2, FK, FT>(hid, fh, fl, tp, lane);
        TileH dh[2], dl[2];
        nlr_split<2, true>(dh, dl, hid);
        nlr_acc_bias<BT>(hb, P.b_d2, h);
        nlr_gemm_x3<BT, 4, 2>(hb, dh, dl, tp, lane);
        TileH hbl[BT];
        nlr_split<BT, false>(hbh, hbl, hb);
        if constexpr (HT > 0) {
            f32x16 hh[HT];
            nlr_acc_bias<HT>(hh, P.b_h1, h);
            nlr_gemm_x3<HT, BT * 2, BT>(hh, hbh, hbl, tp, lane);
            TileH qh[HT], ql[HT];
            nlr_split<HT, true>(qh, ql, hh);
            nlr_acc_bias<1>(lo, P.b_h2, h);
            nlr_gemm_x3<1, HT * 2, HT>(lo, qh, ql, tp, lane);
        }
    }
    if (h == 0 && valid) {
        const float x = hb[0][0] + P.density_bias;
        P.density[sample] = x > 20.0f ? x : log1pf(expf(x));
    }
    // ---- semantic / intensity outputs: rows [0,K) logits -> softmax, row int_row -> intensity
    if constexpr (HT > 0) {
        if (P.K > 0) {  // softmax over rows [0,K) of this column, split over the two lane halves
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nlr_row(r, h) < (int)P.K) mx = fmaxf(mx, lo[0][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float e[16], s = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                e[r] = 0.0f;
                if (nlr_row(r, h) < (int)P.K) {
                    e[r] = expf(lo[0][r] - mx);
                    s += e[r];
                }
            }
            s += __shfl_xor(s, 32, 64);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (nlr_row(r, h) < (int)P.K) P.sem[(size_t)sample * P.K + nlr_row(r, h)] = e[r] / s;
            }
        }
        if (P.inten && valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nlr_row(r, h) == (int)P.int_row) P.inten[sample] = lo[0][r];
        }
    }
    if (P.rgb == nullptr) return;  // density/semantic/intensity only (uniform for the whole grid)

    // ---- view MLP
    const uint32_t ray = sc / P.S;
    const float *rb0 = P.raybias + (size_t)ray * 2 * (WT * 32);
    const float *rb1 = rb0 + WT * 32;
    f32x16 acc[WT];
    f32x16 out1[1];
    if constexpr (!VIEW_F32) {
        nlr_acc_bias<WT>(acc, rb0, h);
        nlr_gemm_bf16<WT, BT * 2, BT>(acc, hbh, tp, lane);
        TileH x[WT];
        nlr_pack<WT, true>(x, acc);
        nlr_acc_bias<WT>(acc, rb1, h);
        nlr_gemm_bf16<WT, WT * 2, WT>(acc, x, tp, lane);
        nlr_gemm_bf16<WT, BT * 2, BT>(acc, hbh, tp, lane);
        nlr_pack<WT, true>(x, acc);
        for (uint32_t l = 2; l < P.depth; ++l) {
            nlr_acc_bias<WT>(acc, P.b_vl + (size_t)(l - 2) * (WT * 32), h);
            nlr_gemm_bf16<WT, WT * 2, WT>(acc, x, tp, lane);
            nlr_pack<WT, true>(x, acc);
        }
        nlr_acc_bias<1>(out1, P.b_rgb, h);
        nlr_gemm_bf16<1, WT * 2, WT>(out1, x, tp, lane);
    } else {
        nlr_acc_bias<WT>(acc, rb0, h);
        nlr_gemm_f32<WT, BT * 4, BT>(acc, hb, tp, lane);
        f32x16 x[WT];
#pragma unroll
        for (int t = 0; t < WT; ++t) x[t] = acc[t];
        nlr_relu<WT>(x);
        nlr_acc_bias<WT>(acc, rb1, h);
        nlr_gemm_f32<WT, WT * 4, WT>(acc, x, tp, lane);
        nlr_gemm_f32<WT, BT * 4, BT>(acc, hb, tp, lane);
#pragma unroll
        for (int t = 0; t < WT; ++t) x[t] = acc[t];
        nlr_relu<WT>(x);
        for (uint32_t l = 2; l < P.depth; ++l) {
            nlr_acc_bias<WT>(acc, P.b_vl + (size_t)(l - 2) * (WT * 32), h);
            nlr_gemm_f32<WT, WT * 4, WT>(acc, x, tp, lane);
#pragma unroll
            for (int t = 0; t < WT; ++t) x[t] = acc[t];
            nlr_relu<WT>(x);
        }
        nlr_acc_bias<1>(out1, P.b_rgb, h);
        nlr_gemm_f32<1, WT * 4, WT>(out1, x, tp, lane);
    }
    if (h == 0 && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float z = P.rgb_premul * out1[0][c] + P.rgb_bias;
            const float sg = 1.0f / (1.0f + expf(-z));
            P.rgb[(size_t)sample * 3 + c] = sg * (1.0f + 2.0f * P.rgb_padding) - P.rgb_padding;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-ray bias of view layers 0 and 1: b_l + W_l[:, dir columns] . pos_enc(viewdir)
// ---------------------------------------------------------------------------------------------

// One 64-thread workgroup per ray: the 3+6*deg encoding is computed once into LDS, then the W outputs of both
// layers are strided over the lanes (weights are read row-contiguously, 27 floats per row).
__global__ void __launch_bounds__(64) nlr_dirbias_kernel(DirBiasParams P) {
    __shared__ float e[64];
    const uint32_t ray = blockIdx.x;
    const int lane = threadIdx.x;
    // coord.py:199-210: [x, sin(2^k x), sin(2^k x + pi/2)], k-major
    if (lane < (int)P.E) {
        float val;
        if (lane < 3) {
            val = P.viewdirs[(size_t)ray * 3 + lane];
        } else {
            const int q = lane - 3;
            const int half = q >= (int)(3 * P.deg);
            const int r = half ? q - 3 * P.deg : q;
            const int k = r / 3, c = r - 3 * k;
            const float sx = P.viewdirs[(size_t)ray * 3 + c] * (float)(1u << k);
            val = half ? sinf(sx + 1.57079637050628662f) : sinf(sx);
        }
        e[lane] = val;
    }
    __syncthreads();
    for (uint32_t j = lane; j < P.W; j += 64) {
        float a0 = P.b0[j], a1 = P.b1[j];
        const float *w0 = P.wd0 + (size_t)j * P.E, *w1 = P.wd1 + (size_t)j * P.E;
        for (uint32_t i = 0; i < P.E; ++i) {
            a0 = fmaf(w0[i], e[i], a0);
            a1 = fmaf(w1[i], e[i], a1);
        }
        P.out[((size_t)ray * 2 + 0) * P.W + j] = a0;
        P.out[((size_t)ray * 2 + 1) * P.W + j] = a1;
    }
}

int nlr_launch_dirbias(const DirBiasParams &P, hipStream_t st) {
    NLR_CHECK_ARG(P.E <= 64, "dir encoding size %u > 64 (deg_view too large)", P.E);
    hipLaunchKernelGGL(nlr_dirbias_kernel, dim3(P.N), dim3(64), 0, st, P);
    NLR_LAUNCH_CHECK("nlr_dirbias_kernel");
    return NLR_OK;
}

// Supported shapes are instantiated explicitly; everything else is reported, not silently emulated.
int nlr_launch_mlp(const MlpParams &P, uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, hipStream_t st) {
    NLR_CHECK_ARG(P.M > 0, "mlp: no samples");
    NLR_CHECK_ARG(P.tape && P.tape_chunks > 0, "mlp: weight tape missing");
    const uint32_t FG = (P.F + 7) / 8;
    dim3 grid((P.M + 127) / 128), block(256);
#define NLR_MLP(WT, BT, FGv, HTv)                                                                                      \
    do {                                                                                                               \
        if (prec == NLR_PREC_F32) hipLaunchKernelGGL((nlr_mlp_kernel<WT, BT, FGv, HTv, NLR_PREC_F32>), grid, block, 0, st, P);        \
        else if (prec == NLR_PREC_MIXED) hipLaunchKernelGGL((nlr_mlp_kernel<WT, BT, FGv, HTv, NLR_PREC_MIXED>), grid, block, 0, st, P); \
        else hipLaunchKernelGGL((nlr_mlp_kernel<WT, BT, FGv, HTv, NLR_PREC_FAST>), grid, block, 0, st, P);            \
        NLR_LAUNCH_CHECK("nlr_mlp_kernel");                                                                            \
        return NLR_OK;                                                                                                 \
    } while (0)
    if (WB == 256 && FG == 5) {
        if (W == 256 && HT == 4) NLR_MLP(8, 8, 5, 4);
        if (W == 256 && HT == 2) NLR_MLP(8, 8, 5, 2);
        if (W == 256 && HT == 0) NLR_MLP(8, 8, 5, 0);
        if (W == 128 && HT == 4) NLR_MLP(4, 8, 5, 4);
        if (W == 128 && HT == 2) NLR_MLP(4, 8, 5, 2);
        if (W == 128 && HT == 0) NLR_MLP(4, 8, 5, 0);
    }
#undef NLR_MLP
    NLR_FAIL(NLR_ERR_UNSUPPORTED,
             "NerfMLP shape (width %u, bottleneck %u, %u grid features, %u head tiles) has no fused kernel instance; "
             "instantiated: width in {128,256}, bottleneck 256, 40 grid features",
             W, WB, P.F, HT);
}
