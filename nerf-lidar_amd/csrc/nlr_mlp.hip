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
//   * the 27 direction-encoding features are computed once per ray by a small pre-kernel and ride through view
//     layers 0 and 1 as one extra zero-padded 32-feature input tile, so every GEMM has K % 32 == 0;
//   * nothing but the weight tape is read from global memory after the prologue: all biases sit in LDS.
//   * precision: layers whose error reaches depth / semantic argmax / intensity (density trunk, heads)
//     use the exact-f32 MFMA (v_mfma_f32_32x32x2_f32); the view MLP (rgb only, 92 % of the MACs) uses
//     bf16 MFMA (v_mfma_f32_32x32x16_bf16) with f32 accumulation.  NLR_PREC_F32 runs everything in f32.
#include "nlr_kernels.h"


// row of accumulator register r for lane half h inside a 32-row tile
__device__ __forceinline__ int nlr_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// accumulator init = bias[row] broadcast over the sample columns; `bias` points into the LDS copy
template <int OT>
__device__ __forceinline__ void nlr_acc_bias(f32x16 (&acc)[OT], const float *bias, int h) {
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + o * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][q * 4 + e] = v[e];
        }
}

// ---- weight tape: global -> registers -> LDS (triple buffered), shared by the 4 waves of a workgroup -----------
// A chunk is 32 KiB = 32 fragments of 1 KiB (one fragment = the A operand of one MFMA for all 64 lanes).  Every
// wave needs every fragment (each wave owns 32 samples and all output features), so staging through LDS cuts the
// L2 -> CU weight traffic 4x against per-wave global loads and puts the fragment reads on ds_read_b128.
// Schedule inside chunk c (f = fragment position, all positions are compile-time after unrolling):
//   f = 20  the 256 threads write chunk c+1 (in registers since chunk c-1, f = 22) into LDS buffer (c+1)%3
//   f = 21  __syncthreads(): chunk c+1 is visible to every wave
//   f = 22  global loads of chunk c+2 are issued (about 30 MFMAs = 900+ cycles of cover until they are needed)
//   every f: the fragment f+8 is requested into an 8-deep register ring right after fragment f is consumed; from
//            f = 24 on these requests run into chunk c+1, so no LDS latency is exposed at a chunk boundary.
// Three buffers: the write at (20, c) lands in the buffer of chunk c-2, whose last read (31, c-2) lies before the
// barrier (21, c-1) that every wave has passed; with two buffers it would race with slow waves still in chunk c-1.
#define NLR_CHUNK_FRAGS 32                       // fragments (1 KiB each) per chunk
#define NLR_CHUNK_SLOTS (NLR_CHUNK_FRAGS * 64)   // uint4 slots per chunk
#define NLR_CHUNK_LOADS (NLR_CHUNK_SLOTS / 256)  // 16-byte loads per thread per chunk
#define NLR_NBUF 3
#define NLR_PF 8                                 // fragment read-ahead (register ring)
struct Tape {
    const uint4 *__restrict__ base;
    uint4 *lds;  // [NLR_NBUF][NLR_CHUNK_SLOTS]
    uint4 nxt[NLR_CHUNK_LOADS];
    uint4 ring[NLR_PF];
    int cur, total, tid, lane;
    __device__ __forceinline__ uint4 *buf(int c) const { return lds + (c % NLR_NBUF) * NLR_CHUNK_SLOTS; }
    __device__ __forceinline__ void load(int c) {
        const uint4 *p = base + (size_t)c * NLR_CHUNK_SLOTS + tid;
#pragma unroll
        for (int r = 0; r < NLR_CHUNK_LOADS; ++r) nxt[r] = p[r * 256];
    }
    __device__ __forceinline__ void store(int c) {
        uint4 *q = buf(c) + tid;
#pragma unroll
        for (int r = 0; r < NLR_CHUNK_LOADS; ++r) q[r * 256] = nxt[r];
    }
    __device__ __forceinline__ void prologue() {
        cur = 0;
        load(0);
        store(0);
        if (total > 1) load(1);
        __syncthreads();
#pragma unroll
        for (int f = 0; f < NLR_PF; ++f) ring[f] = buf(0)[f * 64 + lane];
    }
    // bookkeeping at fragment position F of the current chunk; returns the fragment (raw 16 bytes per lane)
    template <int F>
    __device__ __forceinline__ uint4 step() {
        if (F == 20 && cur + 1 < total) store(cur + 1);
        if (F == 21) __syncthreads();
        if (F == 22 && cur + 2 < total) load(cur + 2);
        const uint4 a = ring[F % NLR_PF];
        if (F + NLR_PF < NLR_CHUNK_FRAGS) {
            ring[F % NLR_PF] = buf(cur)[(F + NLR_PF) * 64 + lane];
        } else if (cur + 1 < total) {
            ring[F % NLR_PF] = buf(cur + 1)[(F + NLR_PF - NLR_CHUNK_FRAGS) * 64 + lane];
        }
        if (F == NLR_CHUNK_FRAGS - 1) ++cur;
        return a;
    }
};

template <typename T>
__device__ __forceinline__ T nlr_as(const uint4 &v) {
    return __builtin_bit_cast(T, v);
}

template <int OT, int KG, int KT, int CH, int F>
__device__ __forceinline__ void nlr_f32_frag(f32x16 (&acc)[OT], const f32x16 (&in)[KT], Tape &tp) {
    const f32x4 a = nlr_as<f32x4>(tp.step<F>());
    constexpr int idx = CH * NLR_CHUNK_FRAGS + F;
    if constexpr (idx < KG * OT) {
        constexpr int g = idx / OT, o = idx % OT;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], in[g >> 2][(g & 3) * 4 + e], acc[o], 0, 0, 0);
    }
    if constexpr (F + 1 < NLR_CHUNK_FRAGS) nlr_f32_frag<OT, KG, KT, CH, F + 1>(acc, in, tp);
}
template <int OT, int KG, int KT, int CH>
__device__ __forceinline__ void nlr_f32_chunk(f32x16 (&acc)[OT], const f32x16 (&in)[KT], Tape &tp) {
    nlr_f32_frag<OT, KG, KT, CH, 0>(acc, in, tp);
    if constexpr ((CH + 1) * NLR_CHUNK_FRAGS < KG * OT) nlr_f32_chunk<OT, KG, KT, CH + 1>(acc, in, tp);
}
// acc[o] += W[o-tile, :] . in   on the exact-f32 MFMA.  KG = number of 8-feature k-groups (one fragment = 4 k-steps).
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_f32(f32x16 (&acc)[OT], const f32x16 (&in)[KT], Tape &tp, int) {
    static_assert(KG <= KT * 4, "k-groups exceed the input tiles");
    nlr_f32_chunk<OT, KG, KT, 0>(acc, in, tp);
}

template <int OT, int KG, int KT, int CH, int F>
__device__ __forceinline__ void nlr_bf16_frag(f32x16 (&acc)[OT], const TileH (&in)[KT], Tape &tp) {
    const bf16x8 a = nlr_as<bf16x8>(tp.step<F>());
    constexpr int idx = CH * NLR_CHUNK_FRAGS + F;
    if constexpr (idx < KG * OT) {
        constexpr int g = idx / OT, o = idx % OT;
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[g >> 1].f[g & 1], acc[o], 0, 0, 0);
    }
    if constexpr (F + 1 < NLR_CHUNK_FRAGS) nlr_bf16_frag<OT, KG, KT, CH, F + 1>(acc, in, tp);
}
template <int OT, int KG, int KT, int CH>
__device__ __forceinline__ void nlr_bf16_chunk(f32x16 (&acc)[OT], const TileH (&in)[KT], Tape &tp) {
    nlr_bf16_frag<OT, KG, KT, CH, 0>(acc, in, tp);
    if constexpr ((CH + 1) * NLR_CHUNK_FRAGS < KG * OT) nlr_bf16_chunk<OT, KG, KT, CH + 1>(acc, in, tp);
}
// acc[o] += W[o-tile, :] . in   on the bf16 MFMA.  KG = number of 16-feature k-steps (= 2 per input tile).
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_bf16(f32x16 (&acc)[OT], const TileH (&in)[KT], Tape &tp, int) {
    static_assert(KG <= KT * 2, "k-steps exceed the input tiles");
    nlr_bf16_chunk<OT, KG, KT, 0>(acc, in, tp);
}

// Split-bf16 ("bf16x3"): W = Wh + Wl, x = xh + xl (each part bf16), W.x ~= Wh.xh + Wh.xl + Wl.xh with f32
// accumulation: 16 mantissa bits per operand (relative error ~2^-16) at 3/16 of the exact-f32 MFMA cost.
// Fragments come in (hi, lo) pairs.
template <int OT, int KG, int KT, int CH, int PR>
__device__ __forceinline__ void nlr_x3_pair(f32x16 (&acc)[OT], const TileH (&inh)[KT], const TileH (&inl)[KT], Tape &tp) {
    const bf16x8 ah = nlr_as<bf16x8>(tp.step<2 * PR>());
    const bf16x8 al = nlr_as<bf16x8>(tp.step<2 * PR + 1>());
    constexpr int idx = CH * (NLR_CHUNK_FRAGS / 2) + PR;
    if constexpr (idx < KG * OT) {
        constexpr int g = idx / OT, o = idx % OT;
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, inh[g >> 1].f[g & 1], acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, inl[g >> 1].f[g & 1], acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, inh[g >> 1].f[g & 1], acc[o], 0, 0, 0);
    }
    if constexpr (PR + 1 < NLR_CHUNK_FRAGS / 2) nlr_x3_pair<OT, KG, KT, CH, PR + 1>(acc, inh, inl, tp);
}
template <int OT, int KG, int KT, int CH>
__device__ __forceinline__ void nlr_x3_chunk(f32x16 (&acc)[OT], const TileH (&inh)[KT], const TileH (&inl)[KT], Tape &tp) {
    nlr_x3_pair<OT, KG, KT, CH, 0>(acc, inh, inl, tp);
    if constexpr ((CH + 1) * (NLR_CHUNK_FRAGS / 2) < KG * OT) nlr_x3_chunk<OT, KG, KT, CH + 1>(acc, inh, inl, tp);
}
template <int OT, int KG, int KT>
__device__ __forceinline__ void nlr_gemm_x3(f32x16 (&acc)[OT], const TileH (&inh)[KT], const TileH (&inl)[KT], Tape &tp, int) {
    static_assert(KG <= KT * 2, "k-steps exceed the input tiles");
    nlr_x3_chunk<OT, KG, KT, 0>(acc, inh, inl, tp);
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
#define NLR_BIAS_MAX 4096  // floats of LDS reserved for the bias block (16 KiB)
template <int WT, int BT, int FG, int HT, int PREC>
__global__ void __launch_bounds__(256, 1) nlr_mlp_kernel(MlpParams P) {
    __shared__ __align__(16) uint4 lds_tape[NLR_NBUF * NLR_CHUNK_SLOTS];
    __shared__ __align__(16) float lds_bias[NLR_BIAS_MAX];
    constexpr bool VIEW_F32 = (PREC == NLR_PREC_F32);
    constexpr bool X3 = (PREC == NLR_PREC_FAST);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const uint32_t sample = (blockIdx.x * 4 + wave) * 32 + col;
    const bool valid = sample < P.M;
    const uint32_t sc = valid ? sample : P.M - 1;
    const uint32_t ray = sc / P.S;
    constexpr int FT = (FG + 3) / 4;
    // bias block offsets (floats)
    constexpr int OB_D0 = 0, OB_D2 = 64, OB_H1 = OB_D2 + BT * 32, OB_H2 = OB_H1 + HT * 32, OB_V0 = OB_H2 + 32;
    constexpr int OB_V1 = OB_V0 + WT * 32, OB_VL = OB_V1 + WT * 32;

    // ---- everything that comes from global memory besides the weight tape is requested up front: the kernel runs
    // one wave per SIMD, so a load in the middle of the chain would be pure exposed latency.
    f32x4 fv[FG];
    {
        const float *fp = P.feat + (size_t)sc * P.F;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            const uint32_t f0 = 8 * g + 4 * h;
            fv[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (f0 + 4 <= P.F) fv[g] = *reinterpret_cast<const f32x4 *>(fp + f0);
        }
    }
    f32x4 ev[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ev[q] = *reinterpret_cast<const f32x4 *>(P.enc + (size_t)ray * 32 + 8 * q + 4 * h);
    for (uint32_t i = threadIdx.x * 4; i < P.bias_count; i += 1024)
        *reinterpret_cast<f32x4 *>(lds_bias + i) = *reinterpret_cast<const f32x4 *>(P.bias_all + i);

    Tape tp;
    tp.base = P.tape;
    tp.lds = lds_tape;
    tp.total = (int)P.tape_chunks;
    tp.tid = threadIdx.x;
    tp.lane = lane;
    tp.prologue();  // ends with __syncthreads(): the bias block is visible too

    // ---- features -> accumulator-layout tiles (lane half h holds features 8q+4h..+3 of each group)
    f32x16 fin[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[t][r] = 0.0f;
#pragma unroll
    for (int g = 0; g < FG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[g >> 2][(g & 3) * 4 + e] = fv[g][e];
    f32x16 encf[1];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) encf[0][q * 4 + e] = ev[q][e];

    f32x16 hb[BT];
    TileH hbe[VIEW_F32 ? 1 : BT + 1];  // bf16 [bottleneck | dir-enc] tiles for the view MLP (hi part in FAST mode)
    f32x16 lo[1];
    if constexpr (!X3) {
        // ---- density_layer.0 : F -> 64, ReLU
        f32x16 hid[2];
        nlr_acc_bias<2>(hid, lds_bias + OB_D0, h);
        nlr_gemm_f32<2, FG, FT>(hid, fin, tp, lane);
        nlr_relu<2>(hid);
        // ---- density_layer.2 : 64 -> bottleneck (no activation); row 0 is the raw density
        nlr_acc_bias<BT>(hb, lds_bias + OB_D2, h);
        nlr_gemm_f32<BT, 8, 2>(hb, hid, tp, lane);
        if constexpr (HT > 0) {
            f32x16 hh[HT];
            nlr_acc_bias<HT>(hh, lds_bias + OB_H1, h);
            nlr_gemm_f32<HT, BT * 4, BT>(hh, hb, tp, lane);
            nlr_relu<HT>(hh);
            nlr_acc_bias<1>(lo, lds_bias + OB_H2, h);
            nlr_gemm_f32<1, HT * 4, HT>(lo, hh, tp, lane);
        }
    } else {
        constexpr int FK = (FG + 1) / 2;  // 16-feature k-steps covering the grid features
        TileH fh[FT], fl[FT];
        nlr_split<FT, false>(fh, fl, fin);
        f32x16 hid[2];
        nlr_acc_bias<2>(hid, lds_bias + OB_D0, h);
        nlr_gemm_x3<2, FK, FT>(hid, fh, fl, tp, lane);
        TileH dh[2], dl[2];
        nlr_split<2, true>(dh, dl, hid);
        nlr_acc_bias<BT>(hb, lds_bias + OB_D2, h);
        nlr_gemm_x3<BT, 4, 2>(hb, dh, dl, tp, lane);
        if constexpr (HT > 0) {
            TileH hbh[BT], hbl[BT];
            nlr_split<BT, false>(hbh, hbl, hb);
            f32x16 hh[HT];
            nlr_acc_bias<HT>(hh, lds_bias + OB_H1, h);
            nlr_gemm_x3<HT, BT * 2, BT>(hh, hbh, hbl, tp, lane);
            TileH qh[HT], ql[HT];
            nlr_split<HT, true>(qh, ql, hh);
            nlr_acc_bias<1>(lo, lds_bias + OB_H2, h);
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

    // ---- view MLP.  Layer 0 input = [bottleneck | enc]; layer 1 input = [x | bottleneck | enc] (skip concat,
    // models.py:1227-1228); the 27 dir-encoding features ride as one extra zero-padded 32-feature input tile.
    f32x16 acc[WT];
    f32x16 out1[1];
    if constexpr (!VIEW_F32) {
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) hbe[t].f[s2][j] = (__bf16)hb[t][8 * s2 + j];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) hbe[BT].f[s2][j] = (__bf16)encf[0][8 * s2 + j];
        nlr_acc_bias<WT>(acc, lds_bias + OB_V0, h);
        nlr_gemm_bf16<WT, (BT + 1) * 2, BT + 1>(acc, hbe, tp, lane);
        TileH x[WT];
        nlr_pack<WT, true>(x, acc);
        nlr_acc_bias<WT>(acc, lds_bias + OB_V1, h);
        nlr_gemm_bf16<WT, WT * 2, WT>(acc, x, tp, lane);
        nlr_gemm_bf16<WT, (BT + 1) * 2, BT + 1>(acc, hbe, tp, lane);
        nlr_pack<WT, true>(x, acc);
        for (uint32_t l = 2; l < P.depth; ++l) {
            nlr_acc_bias<WT>(acc, lds_bias + OB_VL + (l - 2) * (WT * 32), h);
            nlr_gemm_bf16<WT, WT * 2, WT>(acc, x, tp, lane);
            nlr_pack<WT, true>(x, acc);
        }
        nlr_acc_bias<1>(out1, lds_bias + OB_VL + (P.depth - 2) * (WT * 32), h);
        nlr_gemm_bf16<1, WT * 2, WT>(out1, x, tp, lane);
    } else {
        f32x16 hbf[BT + 1];
#pragma unroll
        for (int t = 0; t < BT; ++t) hbf[t] = hb[t];
        hbf[BT] = encf[0];
        nlr_acc_bias<WT>(acc, lds_bias + OB_V0, h);
        nlr_gemm_f32<WT, (BT + 1) * 4, BT + 1>(acc, hbf, tp, lane);
        f32x16 x[WT];
#pragma unroll
        for (int t = 0; t < WT; ++t) x[t] = acc[t];
        nlr_relu<WT>(x);
        nlr_acc_bias<WT>(acc, lds_bias + OB_V1, h);
        nlr_gemm_f32<WT, WT * 4, WT>(acc, x, tp, lane);
        nlr_gemm_f32<WT, (BT + 1) * 4, BT + 1>(acc, hbf, tp, lane);
#pragma unroll
        for (int t = 0; t < WT; ++t) x[t] = acc[t];
        nlr_relu<WT>(x);
        for (uint32_t l = 2; l < P.depth; ++l) {
            nlr_acc_bias<WT>(acc, lds_bias + OB_VL + (l - 2) * (WT * 32), h);
            nlr_gemm_f32<WT, WT * 4, WT>(acc, x, tp, lane);
#pragma unroll
            for (int t = 0; t < WT; ++t) x[t] = acc[t];
            nlr_relu<WT>(x);
        }
        nlr_acc_bias<1>(out1, lds_bias + OB_VL + (P.depth - 2) * (WT * 32), h);
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

// Supported shapes are instantiated explicitly; everything else is reported, not silently emulated.
int nlr_launch_mlp(const MlpParams &P, uint32_t W, uint32_t WB, uint32_t HT, uint32_t prec, hipStream_t st) {
    NLR_CHECK_ARG(P.M > 0, "mlp: no samples");
    NLR_CHECK_ARG(P.tape && P.tape_chunks > 0, "mlp: weight tape missing");
    NLR_CHECK_ARG(P.bias_all && P.bias_count <= NLR_BIAS_MAX && P.bias_count % 4 == 0, "mlp: bias block missing or > %d floats", NLR_BIAS_MAX);
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
