// Fused NerfMLP forward-with-saved-activations and backward for training (scope row f-3).
//
// Replaces, between the hash-grid features and the per-sample heads, what autograd does for the reference's Linear stack in
// `loss.backward()` (Z/train.py:272-281,459 through ZI/models.py:1116-1251): density_layer, sem_layer, intensity_layer,
// lin_second_stage_0..D-1 (+ skip concat), rgb_layer, their ReLUs, softplus / softmax / sigmoid.
//
// Same machinery as the inference kernel (nlr_mlp_kernel.h): transposed chain on v_mfma_f32_16x16x32_bf16, activations and
// activation GRADIENTS in registers as B operands, weights as 1 KiB A fragments streamed from a tape through the LDS ring.
//   forward  tape: D0 D2 [H1 H2] V0 V1 L2..L(D-1) RGB, plain bf16 (mixed-precision training), one wave = 32 samples (2 column
//            tiles), every layer's output also stored row-major in `acts` [M, act_w] bf16 (inputs of the weight gradients and the
//            ReLU masks of the backward)
//   backward tape: RGB^T L(D-1)^T..L2^T V1a^T H2^T BIG D2^T D0^T with BIG = [V1 skip^T | V0^T | H1^T | e_0]: the gradient of the
//            bottleneck collects its four sources (skip concat, view layer 0, heads, raw density) in ONE accumulator chain.
//            Every layer's pre-activation gradient is stored row-major in `gacts`; the gradient of the features goes out in f32.
// Weight gradients are plain GEMMs over the saved tensors (dW_l = gacts_l^T . acts_{l-1}, M-long reductions): library GEMMs on the
// host side (nerflidar_hip/training.py), as are the bias sums.  The tapes are re-packed on the device from the flat parameter
// buffer before every step (nlr_train_pack: one gather through an index map built once).
#include <vector>

#include "nlr_mlp_kernel.h"

struct TrainParams {
    uint32_t M, S, F, depth, K, int_row, act_w;
    const float *feat;   // [M, F] f32
    const float *enc;    // [N, 32]
    const uint4 *tape;
    uint32_t tape_chunks;
    const float *bias_all;
    uint32_t bias_count;
    float density_bias, rgb_premul, rgb_bias, rgb_padding;
    float *density, *rgb, *sem, *inten;  // forward outputs (backward: inputs): [M], [3,M], [K,M], [M]
    __bf16 *acts;                        // [M, act_w]
    const float *g_density, *g_rgb, *g_sem, *g_inten;  // upstream gradients, same layouts; any may be null
    __bf16 *gacts;                       // [M, act_w + 64]
    float *d_feat;                       // [M, F]
};

// row-major [M, ld] bf16 <-> B-operand tiles: lane (col, q) of column tile n holds features c0 + 4q .. +3 (elements 0..3) and
// c0 + 16 + 4q .. +3 (elements 4..7) of sample s0 + 16n + col: two 8-byte accesses per tile
__device__ __forceinline__ void nlr_store_bt(__bf16 *base, uint32_t ld, uint32_t s0, uint32_t M, int col, int q, uint32_t c0, const BT<2> &t) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint32_t smp = s0 + 16 * n + col;
        if (smp < M) {
            const uint4 v = __builtin_bit_cast(uint4, t.n[n]);
            __bf16 *p = base + (size_t)smp * ld + c0 + 4 * q;
            *reinterpret_cast<uint2 *>(p) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2 *>(p + 16) = make_uint2(v.z, v.w);
        }
    }
}
__device__ __forceinline__ void nlr_load_bt(const __bf16 *base, uint32_t ld, uint32_t s0, uint32_t M, int col, int q, uint32_t c0, BT<2> &t) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint32_t smp = s0 + 16 * n + col;
        const uint32_t sc = smp < M ? smp : M - 1;
        const __bf16 *p = base + (size_t)sc * ld + c0 + 4 * q;
        const uint2 a = *reinterpret_cast<const uint2 *>(p), b = *reinterpret_cast<const uint2 *>(p + 16);
        t.n[n] = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
    }
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// gradient piece P (as nlr_pack_piece) masked by the saved post-ReLU activation: x > 0 <=> its bf16 bits are non-zero
template <int P>
__device__ __forceinline__ void nlr_pack_masked(BT<2> &dst, const Unit<2> &src, const BT<2> &act) {
    constexpr int n = P >> 2, jb = (P >> 1) & 1, pr = P & 1;
    const f32x2 x = {src.a[jb][n][2 * pr], src.a[jb][n][2 * pr + 1]};
    const bf16x2 v = __builtin_convertvector(x, bf16x2);
    const u32x4 av = __builtin_bit_cast(u32x4, act.n[n]);
    const uint32_t ab = av[2 * jb + pr];
    const uint32_t m = ((ab & 0xffffu) ? 0xffffu : 0u) | ((ab & 0xffff0000u) ? 0xffff0000u : 0u);
    const uint32_t r = __builtin_bit_cast(uint32_t, v) & m;
    const bf16x2 o = __builtin_bit_cast(bf16x2, r);
    dst.n[n][4 * jb + 2 * pr] = o[0];
    dst.n[n][4 * jb + 2 * pr + 1] = o[1];
}

// WT = view width / 32, BW = bottleneck / 32, FT = ceil(F / 32), HT = head hidden units of 32 (0, 2, 4)
template <int WT, int BW, int FT, int HT, bool BWD>
__global__ void __launch_bounds__(256, 1) nlr_mlp_train_kernel(TrainParams P) {
    __shared__ __align__(16) uint4 lds_tape[NLR_NBUF * NLR_CHUNK_SLOTS];
    __shared__ __align__(16) float lds_bias[NLR_BIAS_MAX];
    __shared__ uint32_t lds_sig;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    constexpr int HTA = HT > 0 ? HT : 1;
    constexpr int OB_D0 = 0, OB_D2 = 64, OB_H1 = OB_D2 + BW * 32, OB_H2 = OB_H1 + HT * 32, OB_V0 = OB_H2 + 32;
    constexpr int OB_V1 = OB_V0 + WT * 32, OB_VL = OB_V1 + WT * 32;
    // columns of acts / gacts
    constexpr uint32_t C_HID = 0, C_HBE = 64, C_Q = C_HBE + BW * 32, C_X = C_Q + HT * 32;
    const uint32_t W = WT * 32, ld = P.act_w, gld = P.act_w + 64, C_LO = P.act_w, C_O = P.act_w + 32;
    // forward program (fragments)
    constexpr int FR_D0 = 2 * FT * 2, FR_D2 = BW * 2 * 2, FR_H1 = HT * BW * 2, FR_H2 = HT > 0 ? HT * 2 : 0;
    constexpr int FR_T = FR_D0 + FR_D2 + FR_H1 + FR_H2;
    constexpr int FR_V0 = WT * (BW + 1) * 2, FR_V1 = WT * (WT + BW + 1) * 2, FR_HL = WT * WT * 2, FR_RGB = WT;
    static_assert(FR_HL % NLR_CHUNK_FRAGS == 0, "hidden view layers must cover whole chunks");
    constexpr int FF_HID = FR_T + FR_V0 + FR_V1, FF_END = FF_HID + FR_RGB;
    // backward program: RGB^T (WT units x 1 k-block), hidden^T, V1a^T (WT x WT), H2^T (HT x 1), BIG (BW x (2 WT + HT + 1)), D2^T (2 x BW), D0^T (FT x 2)
    constexpr int BR_RGB = WT * 1 * 2, BR_V1A = WT * WT * 2, BR_H2 = HT * 1 * 2, BR_BIG = BW * (2 * WT + HT + 1) * 2, BR_D2 = 2 * BW * 2,
                  BR_D0 = FT * 2 * 2;
    constexpr int BF_V1A = BR_RGB, BF_H2 = BF_V1A + BR_V1A, BF_BIG = BF_H2 + BR_H2, BF_D2 = BF_BIG + BR_BIG, BF_D0 = BF_D2 + BR_D2,
                  BF_END = BF_D0 + BR_D0;  // (+ hidden layers between RGB^T and V1a^T: whole chunks)

    const uint32_t ntiles = (P.M + 127) / 128;
    if (!BWD)
        for (uint32_t i = threadIdx.x * 4; i < P.bias_count; i += 1024)
            *reinterpret_cast<f32x4 *>(lds_bias + i) = *reinterpret_cast<const f32x4 *>(P.bias_all + i);
    Tape tp;
    tp.base = P.tape;
    tp.lds = lds_tape;
    tp.sig = &lds_sig;
    tp.sig_addr = (uint32_t)(uintptr_t)(nlr_lptr)&lds_sig;
    tp.total = (int)P.tape_chunks;
    tp.tid = threadIdx.x;
    tp.lane = lane;
    tp.prologue();

    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    auto bias_rows = [&](const float *b, f32x4 (&out)[2]) {
        out[0] = *reinterpret_cast<const f32x4 *>(b + 4 * q);
        out[1] = *reinterpret_cast<const f32x4 *>(b + 16 + 4 * q);
    };
    auto no_bias = [&](auto, f32x4 (&out)[2]) { out[0] = out[1] = zero4; };

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t s0 = tile * 128 + wave * 32;
        if constexpr (!BWD) {
            // ======================================================= forward =======================================================
            BT<2> fb[FT], hbe[BW + 1];
            {
                Unit<2> fin, eu;
#pragma unroll
                for (int t = 0; t < FT; ++t) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const uint32_t smp = s0 + 16 * n + col, sc = smp < P.M ? smp : P.M - 1;
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            const uint32_t f0 = 32 * t + 16 * jb + 4 * q;
                            fin.a[jb][n] = f0 + 4 <= P.F ? *reinterpret_cast<const f32x4 *>(P.feat + (size_t)sc * P.F + f0) : zero4;
                        }
                    }
                    nlr_pack_all<false, 0>(fb[t], fin);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const uint32_t smp = s0 + 16 * n + col, sc = smp < P.M ? smp : P.M - 1;
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) eu.a[jb][n] = *reinterpret_cast<const f32x4 *>(P.enc + (size_t)(sc / P.S) * 32 + 16 * jb + 4 * q);
                }
                nlr_pack_all<false, 0>(hbe[BW], eu);
            }
            BT<2> hidb[2];
            nlr_gemm<2, FT, 2, 2, 1, 0, 9>(
                tp, [&](auto o, f32x4(&b)[2]) { bias_rows(lds_bias + OB_D0 + 32 * decltype(o)::value, b); },
                [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(g)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, fb[G]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc < 8) nlr_pack_piece<true, Pc, 0>(hidb[O], u);
                    else nlr_store_bt(P.acts, ld, s0, P.M, col, q, C_HID + 32 * O, hidb[O]);
                });
            float raw[2] = {0.0f, 0.0f};
            nlr_gemm<BW, 2, 2, 2, 1, FR_D0, 9>(
                tp, [&](auto o, f32x4(&b)[2]) { bias_rows(lds_bias + OB_D2 + 32 * decltype(o)::value, b); },
                [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(g)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, hidb[G]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (O == 0 && Pc == 0) {
                        raw[0] = u.a[0][0][0];
                        raw[1] = u.a[0][1][0];
                    }
                    if constexpr (Pc < 8) nlr_pack_piece<false, Pc, 0>(hbe[O], u);
                    else nlr_store_bt(P.acts, ld, s0, P.M, col, q, C_HBE + 32 * O, hbe[O]);
                });
            Unit<2> lo;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int n = 0; n < 2; ++n) lo.a[jb][n] = zero4;
            if constexpr (HT > 0) {
                BT<2> qb[HTA];
                nlr_gemm<HT, BW, 2, 2, 1, FR_D0 + FR_D2, 9>(
                    tp, [&](auto o, f32x4(&b)[2]) { bias_rows(lds_bias + OB_H1 + 32 * decltype(o)::value, b); },
                    [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        constexpr int G = decltype(g)::value, J = decltype(j)::value;
                        nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, hbe[G]);
                    },
                    [&](auto o, auto p, const Unit<2> &u) {
                        constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                        if constexpr (Pc < 8) nlr_pack_piece<true, Pc, 0>(qb[O], u);
                        else nlr_store_bt(P.acts, ld, s0, P.M, col, q, C_Q + 32 * O, qb[O]);
                    });
                nlr_gemm<1, HT, 2, 2, 1, FR_D0 + FR_D2 + FR_H1, 1>(
                    tp, [&](auto, f32x4(&b)[2]) { bias_rows(lds_bias + OB_H2, b); },
                    [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        constexpr int G = decltype(g)::value, J = decltype(j)::value;
                        nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, qb[G]);
                    },
                    [&](auto, auto, const Unit<2> &u) { lo = u; });
            }
            // per-sample heads (layouts of the inference kernel)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint32_t smp = s0 + 16 * n + col;
                const bool valid = smp < P.M;
                if (q == 0 && valid) {
                    const float x = raw[n] + P.density_bias;
                    P.density[smp] = x > 20.0f ? x : log1pf(expf(x));
                }
                if constexpr (HT > 0) {
                    if (P.K > 0) {
                        float e[2][4], mx = -INFINITY, s = 0.0f;
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                e[jb][r] = (16 * jb + 4 * q + r) < (int)P.K ? lo.a[jb][n][r] : -INFINITY;
                                mx = fmaxf(mx, e[jb][r]);
                            }
                        mx = nlr_q_max(mx);
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                e[jb][r] = expf(e[jb][r] - mx);
                                s += e[jb][r];
                            }
                        s = nlr_q_sum(s);
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * jb + 4 * q + r;
                                if (valid && row < (int)P.K) P.sem[(size_t)row * P.M + smp] = e[jb][r] / s;
                            }
                    }
                    if (P.inten && valid) {
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (16 * jb + 4 * q + r == (int)P.int_row) P.inten[smp] = lo.a[jb][n][r];
                    }
                }
            }
            // view MLP
            BT<2> x[WT], y[WT];
            nlr_gemm<WT, BW + 1, 2, 2, 1, FR_T, 9>(
                tp, [&](auto o, f32x4(&b)[2]) { bias_rows(lds_bias + OB_V0 + 32 * decltype(o)::value, b); },
                [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(g)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, hbe[G]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc < 8) nlr_pack_piece<true, Pc, 0>(x[O], u);
                    else nlr_store_bt(P.acts, ld, s0, P.M, col, q, C_X + 32 * O, x[O]);
                });
            nlr_gemm<WT, WT + BW + 1, 2, 2, 1, FR_T + FR_V0, 9>(
                tp, [&](auto o, f32x4(&b)[2]) { bias_rows(lds_bias + OB_V1 + 32 * decltype(o)::value, b); },
                [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(g)::value, J = decltype(j)::value;
                    if constexpr (G < WT) nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, x[G]);
                    else nlr_mma_bf16<false, 0>(u.a[J], bj, f0, hbe[G - WT]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc < 8) nlr_pack_piece<true, Pc, 0>(y[O], u);
                    else nlr_store_bt(P.acts, ld, s0, P.M, col, q, C_X + W + 32 * O, y[O]);
                });
            for (uint32_t l = 2; l < P.depth; ++l) {
                const float *bl = lds_bias + OB_VL + (l - 2) * (WT * 32);
                const uint32_t cl = C_X + l * W;
                nlr_gemm<WT, WT, 2, 2, 1, FF_HID, 9>(
                    tp, [&](auto o, f32x4(&b)[2]) { bias_rows(bl + 32 * decltype(o)::value, b); },
                    [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        constexpr int G = decltype(g)::value, J = decltype(j)::value;
                        nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, y[G]);
                    },
                    [&](auto o, auto p, const Unit<2> &u) {
                        constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                        if constexpr (Pc < 8) nlr_pack_piece<true, Pc, 0>(x[O], u);
                        else nlr_store_bt(P.acts, ld, s0, P.M, col, q, cl + 32 * O, x[O]);
                    });
#pragma unroll
                for (int t = 0; t < WT; ++t) y[t] = x[t];
            }
            Unit<2> out1;
            nlr_gemm<1, WT, 1, 2, 1, FF_HID, 1>(
                tp, [&](auto, f32x4(&b)[2]) { bias_rows(lds_bias + OB_VL + (P.depth - 2) * (WT * 32), b); },
                [&](Unit<2> &u, auto g, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(g)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, y[G]);
                },
                [&](auto, auto, const Unit<2> &u) { out1 = u; });
            if (q == 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const uint32_t smp = s0 + 16 * n + col;
                    if (smp < P.M) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float z = P.rgb_premul * out1.a[0][n][c] + P.rgb_bias;
                            const float sg = 1.0f / (1.0f + expf(-z));
                            P.rgb[(size_t)c * P.M + smp] = sg * (1.0f + 2.0f * P.rgb_padding) - P.rgb_padding;
                        }
                    }
                }
            }
            nlr_pad<FF_END % NLR_CHUNK_FRAGS>(tp);
        } else {
            // ======================================================= backward ======================================================
            // d loss / d (rgb_layer output): rgb = s (1 + 2p) - p, s = sigmoid(premul o + bias)
            BT<2> gin;  // one 32-feature k-block of upstream gradient
            {
                Unit<2> du;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int n = 0; n < 2; ++n) du.a[jb][n] = zero4;
                if (q == 0 && P.g_rgb) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const uint32_t smp = s0 + 16 * n + col;
                        if (smp < P.M) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const float s = (P.rgb[(size_t)c * P.M + smp] + P.rgb_padding) / (1.0f + 2.0f * P.rgb_padding);
                                du.a[0][n][c] = P.g_rgb[(size_t)c * P.M + smp] * (1.0f + 2.0f * P.rgb_padding) * (s * (1.0f - s)) * P.rgb_premul;
                            }
                        }
                    }
                }
                nlr_pack_all<false, 0>(gin, du);
                nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_O, gin);
            }
            BT<2> g[WT], h[WT], mk;
            {  // RGB^T: gradient of the last view layer's pre-activation
                const uint32_t cl = C_X + (P.depth - 1) * W;
                nlr_gemm<WT, 1, 2, 2, 1, 0, 9>(
                    tp, no_bias,
                    [&](Unit<2> &u, auto, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        nlr_mma_bf16<true, 0>(u.a[decltype(j)::value], bj, f0, gin);
                    },
                    [&](auto o, auto p, const Unit<2> &u) {
                        constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                        if constexpr (Pc == 0) nlr_load_bt(P.acts, ld, s0, P.M, col, q, cl + 32 * O, mk);
                        if constexpr (Pc < 8) nlr_pack_masked<Pc>(g[O], u, mk);
                        else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, cl + 32 * O, g[O]);
                    });
            }
            for (uint32_t l = P.depth - 1; l >= 2; --l) {  // hidden layers, transposed: d z_{l-1} = mask(x_{l-1}) W_l^T d z_l
                const uint32_t cl = C_X + (l - 1) * W;
                nlr_gemm<WT, WT, 2, 2, 1, BR_RGB, 9>(
                    tp, no_bias,
                    [&](Unit<2> &u, auto gg, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        constexpr int G = decltype(gg)::value, J = decltype(j)::value;
                        nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, g[G]);
                    },
                    [&](auto o, auto p, const Unit<2> &u) {
                        constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                        if constexpr (Pc == 0) nlr_load_bt(P.acts, ld, s0, P.M, col, q, cl + 32 * O, mk);
                        if constexpr (Pc < 8) nlr_pack_masked<Pc>(h[O], u, mk);
                        else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, cl + 32 * O, h[O]);
                    });
#pragma unroll
                for (int t = 0; t < WT; ++t) g[t] = h[t];
            }
            // g = d z_1.  V1a^T: d z_0 = mask(x_0) W1[:, :W]^T d z_1
            nlr_gemm<WT, WT, 2, 2, 1, BF_V1A, 9>(
                tp, no_bias,
                [&](Unit<2> &u, auto gg, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(gg)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, g[G]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc == 0) nlr_load_bt(P.acts, ld, s0, P.M, col, q, C_X + 32 * O, mk);
                    if constexpr (Pc < 8) nlr_pack_masked<Pc>(h[O], u, mk);
                    else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_X + 32 * O, h[O]);
                });
            // heads: softmax backward d l_c = p_c (g_c - sum_k g_k p_k), intensity row: its upstream gradient
            BT<2> gq[HTA], aux;
            {
                Unit<2> dl, da;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int n = 0; n < 2; ++n) dl.a[jb][n] = da.a[jb][n] = zero4;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const uint32_t smp = s0 + 16 * n + col;
                    const bool valid = smp < P.M;
                    if constexpr (HT > 0) {
                        if (P.K > 0 && P.g_sem) {
                            float pr[2][4], gr[2][4], dot = 0.0f;
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * jb + 4 * q + r;
                                    const bool on = valid && row < (int)P.K;
                                    pr[jb][r] = on ? P.sem[(size_t)row * P.M + smp] : 0.0f;
                                    gr[jb][r] = on ? P.g_sem[(size_t)row * P.M + smp] : 0.0f;
                                    dot += pr[jb][r] * gr[jb][r];
                                }
                            dot = nlr_q_sum(dot);
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) dl.a[jb][n][r] = pr[jb][r] * (gr[jb][r] - dot);
                        }
                        if (P.g_inten && valid) {
#pragma unroll
                            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (16 * jb + 4 * q + r == (int)P.int_row) dl.a[jb][n][r] = P.g_inten[smp];
                        }
                    }
                    // raw density: density = softplus(raw + bias) -> d raw = g (1 - exp(-density)); fed to BIG as two bf16 features
                    // (hi + lo) with unit weights on bottleneck row 0
                    if (q == 0 && valid && P.g_density) {
                        const float dr = P.g_density[smp] * (1.0f - expf(-P.density[smp]));
                        const float hi = (float)(__bf16)dr;
                        da.a[0][n][0] = hi;
                        da.a[0][n][1] = dr - hi;
                    }
                }
                nlr_pack_all<false, 0>(gin, dl);
                nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_LO, gin);
                nlr_pack_all<false, 0>(aux, da);
            }
            if constexpr (HT > 0) {  // H2^T: d (head hidden pre-activation)
                nlr_gemm<HT, 1, 2, 2, 1, BF_H2, 9>(
                    tp, no_bias,
                    [&](Unit<2> &u, auto, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                        nlr_mma_bf16<true, 0>(u.a[decltype(j)::value], bj, f0, gin);
                    },
                    [&](auto o, auto p, const Unit<2> &u) {
                        constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                        if constexpr (Pc == 0) nlr_load_bt(P.acts, ld, s0, P.M, col, q, C_Q + 32 * O, mk);
                        if constexpr (Pc < 8) nlr_pack_masked<Pc>(gq[O], u, mk);
                        else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_Q + 32 * O, gq[O]);
                    });
            }
            // BIG: d bottleneck = W1[:, W:W+WB]^T d z_1 + W0[:, :WB]^T d z_0 + H1^T d q + e_0 d raw
            BT<2> gb[BW];
            nlr_gemm<BW, 2 * WT + HT + 1, 2, 2, 1, BF_BIG, 9>(
                tp, no_bias,
                [&](Unit<2> &u, auto gg, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(gg)::value, J = decltype(j)::value;
                    if constexpr (G < WT) nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, g[G]);
                    else if constexpr (G < 2 * WT) nlr_mma_bf16<false, 0>(u.a[J], bj, f0, h[G - WT]);
                    else if constexpr (G < 2 * WT + HT) nlr_mma_bf16<false, 0>(u.a[J], bj, f0, gq[G - 2 * WT]);
                    else nlr_mma_bf16<false, 0>(u.a[J], bj, f0, aux);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc < 8) nlr_pack_piece<false, Pc, 0>(gb[O], u);
                    else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_HBE + 32 * O, gb[O]);
                });
            // D2^T: d (trunk hidden pre-activation)
            BT<2> gh[2];
            nlr_gemm<2, BW, 2, 2, 1, BF_D2, 9>(
                tp, no_bias,
                [&](Unit<2> &u, auto gg, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(gg)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, gb[G]);
                },
                [&](auto o, auto p, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                    if constexpr (Pc == 0) nlr_load_bt(P.acts, ld, s0, P.M, col, q, C_HID + 32 * O, mk);
                    if constexpr (Pc < 8) nlr_pack_masked<Pc>(gh[O], u, mk);
                    else nlr_store_bt(P.gacts, gld, s0, P.M, col, q, C_HID + 32 * O, gh[O]);
                });
            // D0^T: gradient of the grid features, f32 out
            nlr_gemm<FT, 2, 2, 2, 1, BF_D0, 1>(
                tp, no_bias,
                [&](Unit<2> &u, auto gg, auto j, const uint4 &f0, const uint4 &, const f32x4 &bj) {
                    constexpr int G = decltype(gg)::value, J = decltype(j)::value;
                    nlr_mma_bf16<G == 0, 0>(u.a[J], bj, f0, gh[G]);
                },
                [&](auto o, auto, const Unit<2> &u) {
                    constexpr int O = decltype(o)::value;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const uint32_t smp = s0 + 16 * n + col;
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            const uint32_t f0 = 32 * O + 16 * jb + 4 * q;
                            if (smp < P.M && f0 + 4 <= P.F) *reinterpret_cast<f32x4 *>(P.d_feat + (size_t)smp * P.F + f0) = u.a[jb][n];
                        }
                    }
                });
            nlr_pad<BF_END % NLR_CHUNK_FRAGS>(tp);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- device-side tape packing: tape element i (bf16) = flat parameter idx[i] (or 0 when idx[i] < 0); bias block likewise in f32
__global__ void __launch_bounds__(256) nlr_train_pack_kernel(const float *__restrict__ params, const int32_t *__restrict__ idx, uint32_t n,
                                                            __bf16 *__restrict__ tape, const int32_t *__restrict__ bidx, uint32_t nb,
                                                            float *__restrict__ bias) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int32_t k = idx[i];  // -1: structural zero, -2: structural one
        tape[i] = k >= 0 ? (__bf16)params[k] : (k == -2 ? (__bf16)1.0f : (__bf16)0.0f);
    }
    if (i < nb) {
        const int32_t k = bidx[i];
        bias[i] = k >= 0 ? params[k] : 0.0f;
    }
}

// ---- plan: shapes, parameter order, index maps ----------------------------------------------------------------------------------
struct IMat {  // matrix of flat-parameter indices (-1 = structural zero)
    std::vector<int32_t> a;
    uint32_t rows = 0, cols = 0;
    IMat() {}
    IMat(uint32_t r, uint32_t c) : a((size_t)r * c, -1), rows(r), cols(c) {}
    int32_t &at(uint32_t r, uint32_t c) { return a[(size_t)r * cols + c]; }
    int32_t get(uint32_t r, uint32_t c) const { return (r < rows && c < cols) ? a[(size_t)r * cols + c] : -1; }
};
static void tape_add(std::vector<int32_t> &t, const IMat &w, uint32_t out_pad, uint32_t in_pad, uint32_t RH = 2) {
    const uint32_t OT = RH == 2 ? out_pad / 32 : 1, KG = in_pad / 32;
    for (uint32_t o = 0; o < OT; ++o)
        for (uint32_t g = 0; g < KG; ++g)
            for (uint32_t j = 0; j < RH; ++j) {
                const uint32_t R = RH == 2 ? 2 * o + j : o;
                for (uint32_t lane = 0; lane < 64; ++lane)
                    for (uint32_t e = 0; e < 8; ++e) t.push_back(w.get(16 * R + (lane & 15), 32 * g + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3)));
            }
}
static void tape_pad(std::vector<int32_t> &t) { t.resize((t.size() + 16383) / 16384 * 16384, -1); }  // 32 KiB of bf16

struct NlrTrainPlan {
    uint32_t F, W, WB, HT, D, K, int_row, act_w, n_params, cus;
    bool sem, inten;
    float density_bias, rgb_premul, rgb_bias, rgb_padding;
    int32_t *fidx = nullptr, *bidx = nullptr, *biasidx = nullptr;
    uint32_t fn = 0, bn = 0, biasn = 0;
    __bf16 *ftape = nullptr, *btape = nullptr;
    float *bias = nullptr;
    std::vector<uint32_t> offs;  // flat offsets: see nlr_train_param_layout
};

static IMat lin(uint32_t off, uint32_t rows, uint32_t cols) {
    IMat m(rows, cols);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t c = 0; c < cols; ++c) m.at(r, c) = (int32_t)(off + r * cols + c);
    return m;
}
static IMat transpose(const IMat &w) {
    IMat t(w.cols, w.rows);
    for (uint32_t r = 0; r < w.rows; ++r)
        for (uint32_t c = 0; c < w.cols; ++c) t.at(c, r) = w.a[(size_t)r * w.cols + c];
    return t;
}

// Flat parameter order (weights row-major [out, in], then bias), `n_entries` pairs:
//   density_layer.0, density_layer.2, [sem_layer.0, sem_layer.2], [intensity_layer.0, intensity_layer.2], lin_second_stage_0..D-1, rgb_layer
extern "C" int nlr_train_plan_create(uint32_t F, uint32_t W, uint32_t WB, uint32_t D, uint32_t deg_view, uint32_t class_num, int use_semantic,
                                     int use_intensity, float density_bias, float rgb_premultiplier, float rgb_bias, float rgb_padding,
                                     NlrTrainPlan **out, uint32_t *n_params) {
    NLR_CHECK_ARG(out && n_params, "train_plan_create: NULL argument");
    const uint32_t E = 3 + 6 * deg_view;
    NLR_CHECK_ARG(WB == 256 && (W == 128 || W == 256) && F % 4 == 0 && F > 32 && F <= 64 && D >= 2 && D <= 10 && E <= 32 && class_num <= 31,
                  "train_plan_create: unsupported NerfMLP shape (bottleneck 256, view width 128/256, 33..64 grid features, depth 2..10)");
    NlrTrainPlan *p = new NlrTrainPlan();
    p->F = F, p->W = W, p->WB = WB, p->D = D, p->sem = use_semantic != 0, p->inten = use_intensity != 0;
    p->K = p->sem ? class_num : 0;
    p->int_row = p->inten ? p->K : 0xffffffffu;
    p->HT = (p->sem ? 2 : 0) + (p->inten ? 2 : 0);
    p->density_bias = density_bias, p->rgb_premul = rgb_premultiplier, p->rgb_bias = rgb_bias, p->rgb_padding = rgb_padding;
    p->act_w = 64 + WB + p->HT * 32 + D * W;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        delete p;
        NLR_FAIL(NLR_ERR_HIP, "train_plan_create: cannot query the current device");
    }
    p->cus = (uint32_t)cus;
    uint32_t off = 0;
    auto take = [&](uint32_t rows, uint32_t cols, IMat &w, uint32_t &b) {
        w = lin(off, rows, cols);
        off += rows * cols;
        b = off;
        off += rows;
        p->offs.push_back(b - rows * cols);
        p->offs.push_back(b);
    };
    IMat d0, d2, s0, s2, i0, i2, rgbw;
    std::vector<IMat> v(D);
    uint32_t bd0, bd2, bs0 = 0, bs2 = 0, bi0 = 0, bi2 = 0, brgb;
    std::vector<uint32_t> bv(D);
    take(64, F, d0, bd0);
    take(WB, 64, d2, bd2);
    if (p->sem) {
        take(64, WB, s0, bs0);
        take(class_num, 64, s2, bs2);
    }
    if (p->inten) {
        take(64, WB, i0, bi0);
        take(1, 64, i2, bi2);
    }
    const uint32_t in0 = WB + E, in1 = W + in0;
    for (uint32_t l = 0; l < D; ++l) take(W, l == 0 ? in0 : (l == 1 ? in1 : W), v[l], bv[l]);
    take(3, W, rgbw, brgb);
    p->n_params = off;
    // heads as two stacked GEMMs (as the inference path): h1 = [sem0 ; int0], h2 block-diagonal into one 32-row unit
    const uint32_t HH = p->HT * 32;
    IMat h1(HH ? HH : 1, WB), h2(32, HH ? HH : 1);
    std::vector<int32_t> b1(HH, -1), b2(32, -1);
    {
        uint32_t r0 = 0;
        if (p->sem) {
            for (uint32_t r = 0; r < 64; ++r) {
                for (uint32_t c = 0; c < WB; ++c) h1.at(r0 + r, c) = s0.a[(size_t)r * WB + c];
                b1[r0 + r] = (int32_t)(bs0 + r);
            }
            for (uint32_t r = 0; r < class_num; ++r) {
                for (uint32_t c = 0; c < 64; ++c) h2.at(r, r0 + c) = s2.a[(size_t)r * 64 + c];
                b2[r] = (int32_t)(bs2 + r);
            }
            r0 += 64;
        }
        if (p->inten) {
            for (uint32_t r = 0; r < 64; ++r) {
                for (uint32_t c = 0; c < WB; ++c) h1.at(r0 + r, c) = i0.a[(size_t)r * WB + c];
                b1[r0 + r] = (int32_t)(bi0 + r);
            }
            for (uint32_t c = 0; c < 64; ++c) h2.at(p->int_row, r0 + c) = i2.a[c];
            b2[p->int_row] = (int32_t)bi2;
        }
    }
    // view layers with the direction-encoding columns padded to one 32-feature k-block
    auto pad_enc = [&](const IMat &w, uint32_t lead) {  // [.., lead + E] -> [.., lead + 32]
        IMat m(w.rows, lead + 32);
        for (uint32_t r = 0; r < w.rows; ++r)
            for (uint32_t c = 0; c < w.cols; ++c) m.at(r, c) = w.a[(size_t)r * w.cols + c];
        return m;
    };
    const IMat v0p = pad_enc(v[0], WB), v1p = pad_enc(v[1], W + WB);
    // ---- forward tape + bias block
    std::vector<int32_t> ft, bt, bb;
    tape_add(ft, d0, 64, 64);
    tape_add(ft, d2, WB, 64);
    if (HH) {
        tape_add(ft, h1, HH, WB);
        tape_add(ft, h2, 32, HH);
    }
    tape_add(ft, v0p, W, WB + 32);
    tape_add(ft, v1p, W, W + WB + 32);
    for (uint32_t l = 2; l < D; ++l) tape_add(ft, v[l], W, W);
    tape_add(ft, rgbw, 16, W, 1);
    tape_pad(ft);
    auto pushb = [&](uint32_t b, uint32_t n, uint32_t pad) {
        for (uint32_t i = 0; i < pad; ++i) bb.push_back(i < n ? (int32_t)(b + i) : -1);
    };
    pushb(bd0, 64, 64);
    pushb(bd2, WB, WB);
    for (uint32_t i = 0; i < HH; ++i) bb.push_back(b1[i]);
    for (uint32_t i = 0; i < 32; ++i) bb.push_back(HH ? b2[i] : -1);
    for (uint32_t l = 0; l < D; ++l) pushb(bv[l], W, W);
    pushb(brgb, 3, 32);
    // ---- backward tape
    tape_add(bt, transpose(rgbw), W, 32);  // rows = view features, k = the 3 (of 32) rgb rows
    for (uint32_t l = D - 1; l >= 2; --l) tape_add(bt, transpose(v[l]), W, W);
    {
        IMat v1a(W, W);  // d z_0 <- d z_1 through W1[:, :W]
        for (uint32_t r = 0; r < W; ++r)
            for (uint32_t c = 0; c < W; ++c) v1a.at(r, c) = v1p.a[(size_t)c * v1p.cols + r];
        tape_add(bt, v1a, W, W);
    }
    if (HH) tape_add(bt, transpose(h2), HH, 32);
    {
        IMat big(WB, 2 * W + HH + 32);
        for (uint32_t r = 0; r < WB; ++r) {
            for (uint32_t c = 0; c < W; ++c) big.at(r, c) = v1p.a[(size_t)c * v1p.cols + W + r];       // skip concat of layer 1
            for (uint32_t c = 0; c < W; ++c) big.at(r, W + c) = v0p.a[(size_t)c * v0p.cols + r];       // layer 0
            for (uint32_t c = 0; c < HH; ++c) big.at(r, 2 * W + c) = h1.a[(size_t)c * WB + r];         // heads
        }
        big.at(0, 2 * W + HH + 0) = -2;  // unit weights: raw-density gradient (hi, lo) onto bottleneck row 0
        big.at(0, 2 * W + HH + 1) = -2;
        tape_add(bt, big, WB, 2 * W + HH + 32);
    }
    tape_add(bt, transpose(d2), 64, WB);
    {
        IMat d0t(64, 64);
        for (uint32_t r = 0; r < F; ++r)
            for (uint32_t c = 0; c < 64; ++c) d0t.at(r, c) = d0.a[(size_t)c * F + r];
        tape_add(bt, d0t, 64, 64);
    }
    tape_pad(bt);
    p->fn = (uint32_t)ft.size(), p->bn = (uint32_t)bt.size(), p->biasn = (uint32_t)bb.size();
    NLR_CHECK_ARG(p->biasn <= NLR_BIAS_MAX && p->biasn % 4 == 0, "train_plan_create: bias block of %u floats does not fit", p->biasn);
    auto up = [&](const std::vector<int32_t> &h, int32_t **d) -> int {
        NLR_HIP(hipMalloc((void **)d, h.size() * 4));
        NLR_HIP(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        return NLR_OK;
    };
    int rc;
    if ((rc = up(ft, &p->fidx)) || (rc = up(bt, &p->bidx)) || (rc = up(bb, &p->biasidx))) return rc;
    const size_t slack = 3 * 16384;  // the kernel prefetches up to 3 chunks past the end
    NLR_HIP(hipMalloc((void **)&p->ftape, (ft.size() + slack) * 2));
    NLR_HIP(hipMalloc((void **)&p->btape, (bt.size() + slack) * 2));
    NLR_HIP(hipMemset(p->ftape, 0, (ft.size() + slack) * 2));
    NLR_HIP(hipMemset(p->btape, 0, (bt.size() + slack) * 2));
    NLR_HIP(hipMalloc((void **)&p->bias, bb.size() * 4));
    *out = p;
    *n_params = p->n_params;
    return NLR_OK;
}

extern "C" void nlr_train_plan_destroy(NlrTrainPlan *p) {
    if (!p) return;
    (void)hipFree(p->fidx), (void)hipFree(p->bidx), (void)hipFree(p->biasidx), (void)hipFree(p->ftape), (void)hipFree(p->btape), (void)hipFree(p->bias);
    delete p;
}

extern "C" uint32_t nlr_train_act_width(const NlrTrainPlan *p) { return p ? p->act_w : 0; }

// offsets (in floats) of every (weight, bias) pair inside the flat parameter buffer, in the order documented above
extern "C" int nlr_train_param_layout(const NlrTrainPlan *p, uint32_t *offsets, uint32_t capacity) {
    NLR_CHECK_ARG(p && offsets && capacity >= p->offs.size(), "train_param_layout: buffer too small (%zu entries)", p ? p->offs.size() : 0);
    for (size_t i = 0; i < p->offs.size(); ++i) offsets[i] = p->offs[i];
    return (int)p->offs.size();
}

extern "C" int nlr_train_pack(NlrTrainPlan *p, const float *params_dev, void *stream) {
    NLR_CHECK_ARG(p && params_dev, "train_pack: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nlr_train_pack_kernel, dim3((p->fn + 255) / 256), dim3(256), 0, st, params_dev, p->fidx, p->fn, p->ftape, p->biasidx,
                       p->biasn, p->bias);
    hipLaunchKernelGGL(nlr_train_pack_kernel, dim3((p->bn + 255) / 256), dim3(256), 0, st, params_dev, p->bidx, p->bn, p->btape, nullptr, 0u,
                       nullptr);
    NLR_LAUNCH_CHECK("nlr_train_pack_kernel");
    return NLR_OK;
}

template <bool BWD>
static int launch_train(const NlrTrainPlan *p, TrainParams &P, hipStream_t st) {
    const uint32_t ntiles = (P.M + 127) / 128;
    dim3 grid(ntiles < p->cus ? ntiles : p->cus);
#define NLR_TR(wt, ht)                                                                                              \
    if (p->W == wt * 32 && p->HT == ht) {                                                                           \
        hipLaunchKernelGGL((nlr_mlp_train_kernel<wt, 8, 2, ht, BWD>), grid, dim3(256), 0, st, P);                    \
        NLR_LAUNCH_CHECK("nlr_mlp_train_kernel");                                                                    \
        return NLR_OK;                                                                                              \
    }
    NLR_TR(8, 4) NLR_TR(8, 2) NLR_TR(8, 0) NLR_TR(4, 4) NLR_TR(4, 2) NLR_TR(4, 0)
#undef NLR_TR
    NLR_FAIL(NLR_ERR_UNSUPPORTED, "mlp_train: no kernel instance for view width %u with %u head units", p->W, p->HT);
}

static void fill_common(const NlrTrainPlan *p, TrainParams &P, uint32_t M, uint32_t S) {
    memset(&P, 0, sizeof(P));
    P.M = M, P.S = S, P.F = p->F, P.depth = p->D, P.K = p->K, P.int_row = p->int_row, P.act_w = p->act_w;
    P.density_bias = p->density_bias, P.rgb_premul = p->rgb_premul, P.rgb_bias = p->rgb_bias, P.rgb_padding = p->rgb_padding;
}

// features [M, F] f32 row-major, enc [M / S, 32]; outputs as nlr_mlp_level; acts [M, act_w] bf16
extern "C" int nlr_mlp_train_forward(const NlrTrainPlan *p, const float *features, const float *enc, uint32_t M, uint32_t S, float *density,
                                     float *rgb, float *semantic, float *intensity, void *acts, void *stream) {
    NLR_CHECK_ARG(p && features && enc && density && rgb && acts && M > 0 && S > 0, "mlp_train_forward: bad argument");
    NLR_CHECK_ARG((!p->sem || semantic) && (!p->inten || intensity), "mlp_train_forward: head output missing");
    TrainParams P;
    fill_common(p, P, M, S);
    P.feat = features, P.enc = enc;
    P.tape = (const uint4 *)p->ftape, P.tape_chunks = p->fn / 16384;
    P.bias_all = p->bias, P.bias_count = p->biasn;
    P.density = density, P.rgb = rgb, P.sem = semantic, P.inten = p->inten ? intensity : nullptr;
    P.acts = (__bf16 *)acts;
    return launch_train<false>(p, P, (hipStream_t)stream);
}

// upstream gradients in the layouts of the outputs (any may be NULL); gacts [M, act_w + 64] bf16, d_features [M, F] f32
extern "C" int nlr_mlp_train_backward(const NlrTrainPlan *p, uint32_t M, uint32_t S, const float *density, const float *rgb, const float *semantic,
                                      const void *acts, const float *g_density, const float *g_rgb, const float *g_semantic,
                                      const float *g_intensity, void *gacts, float *d_features, void *stream) {
    NLR_CHECK_ARG(p && density && rgb && acts && gacts && d_features && M > 0 && S > 0, "mlp_train_backward: bad argument");
    TrainParams P;
    fill_common(p, P, M, S);
    P.tape = (const uint4 *)p->btape, P.tape_chunks = p->bn / 16384;
    P.density = const_cast<float *>(density), P.rgb = const_cast<float *>(rgb), P.sem = const_cast<float *>(semantic);
    P.acts = (__bf16 *)const_cast<void *>(acts);
    P.g_density = g_density, P.g_rgb = g_rgb, P.g_sem = (p->sem && semantic) ? g_semantic : nullptr, P.g_inten = p->inten ? g_intensity : nullptr;
    P.gacts = (__bf16 *)gacts;
    P.d_feat = d_features;
    return launch_train<true>(p, P, (hipStream_t)stream);
}
