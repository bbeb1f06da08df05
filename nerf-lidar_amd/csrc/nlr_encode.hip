// Multisample cone casting + scene contraction + hash-grid encoding + erf down-weighting, fused.
//
// Replaces, per sampling level (rows a-5..a-9 of the scope table):
//   ZI/render.py:129-168     cast_rays            -> means[N,S,7,3], stds[N,S,7]     (never stored here)
//   ZI/coord.py:51-63,67-100 contract_mean_std    (track_linearize('contract'))
//   ZI/models.py:968-979     /bound, GridEncoder, erf re-weighting, mean over the 7 multisamples
//   gridencoder.cu:87-199    kernel_grid          (8-corner trilinear gather per level)
//   ZI/models.py:887-889,996-997,1116  PropMLP density_layer + softplus (proposal levels only)
// The reference writes means/stds/[B,L*C] raw features to HBM between these steps (for 32x1024 rays
// x 128 samples x 7 that is 29.4 M points x 40 floats = 4.7 GB per sweep); here only the
// 7-sample mean [N*S, L*C] leaves the kernel (NerfMLP level) or only the density (proposal levels).
#include "nlr_kernels.h"

#include <stdlib.h>

#include "nlr_grid_level.h"
#include "nlr_level_fast.h"



__device__ __forceinline__ Gauss nlr_cast_one(const CastParams &cp, uint32_t ray, uint32_t k, uint32_t j, float t0, float t1,
                                              const float *o, const float *d, const float *bx, const float *by, float radius,
                                              float *raw = nullptr) {
    // render.py:147-156
    const float t = t0 + ((t1 - t0) * ((float)j + 0.5f)) / (float)cp.n;
    float cd = cp.cosd[j], sd = cp.sind[j];
    if (cp.rand_deg) {
        const float u = cp.rand_deg[((size_t)ray * cp.S + k) * cp.n + j];
        const float deg = cp.degj[j] + (u * 3.14159274101257324f) * 2.0f;
        cd = cosf(deg);
        sd = sinf(deg);
    }
    const float lx = ((radius * t) * cd) / 2.0f;
    const float ly = ((radius * t) * sd) / 2.0f;
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = fmaf(t, d[c], fmaf(ly, by[c], lx * bx[c])) + o[c];  // render.py:162-164
    const float std = (cp.std_scale * radius) * t;
    // coord.py:51-63
    const float m2 = fmaxf((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2], NLR_EPS);
    float zs = std;
    if (!(m2 <= 1.0f)) {
        const float sq = sqrtf(m2);
        const float sc = (2.0f * sq - 1.0f) / m2;
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = sc * m[c];
        const float a = 2.0f / sq - 1.0f / m2;
        const float det = (1.0f / m2) * (a * a);
        // det ** (1/3) (coord.py:61).  cbrtf (17 VALU instructions) instead of powf(det, 0.33333334f) (164): the two differ by
        // det^(1e-8), i.e. <= 2 ulp, in a value that only scales the erf re-weighting argument
        zs = cbrtf(det) * std;
    }
    Gauss g;
    // models.py:971-973 (bound = 2) then grid.py:162 ((x + 1) / 2)
    g.x0 = (m[0] / 2.0f + 1.0f) / 2.0f;
    g.x1 = (m[1] / 2.0f + 1.0f) / 2.0f;
    g.x2 = (m[2] / 2.0f + 1.0f) / 2.0f;
    g.zs = zs / 2.0f;
    if (raw) {  // means / bound as GridEncoder(bound=1) takes them (models.py:969-973)
        raw[0] = m[0] / 2.0f;
        raw[1] = m[1] / 2.0f;
        raw[2] = m[2] / 2.0f;
    }
    return g;
}

// models.py:976: erf(1 / clamp(sqrt(8 * std^2 * G^2), min=1e-10))
__device__ __forceinline__ float nlr_erf_weight(float zs, float gsize) {
    return erff(1.0f / fmaxf(sqrtf((8.0f * (zs * zs)) * (gsize * gsize)), 1e-10f));
}
// The same weight with the level-independent part hoisted: 1/sqrt(8 std^2 G^2) = inv_s8 / G with
// inv_s8 = 1/sqrt(8 std^2) computed once per multisample (differs from the reference's rounding order by a few
// ulp of the erf ARGUMENT, i.e. <= 3e-7 of the weight; the clamp at 1e-10 maps to an argument cap of 1e10).
__device__ __forceinline__ float nlr_erf_weight_fast(float inv_s8, float inv_g) {
    return erff(fminf(inv_s8 * inv_g, 1e10f));
}

// ---------------------------------------------------------------------------------------------
// NerfMLP level: one thread per (sample, grid level); blockIdx.y = level (level-major dispatch keeps
// one level's table hot in L2 / Infinity Cache).  Writes feat[M, L*C] (row-major, f32).
// ---------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ void __launch_bounds__(256) nlr_encode_kernel(CastParams cp, GridParams gp, int re_weights, float *__restrict__ feat) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = cp.N * cp.S;
    if (m >= M) return;
    const uint32_t level = blockIdx.y;
    const uint32_t ray = m / cp.S, k = m - ray * cp.S;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float radius = cp.radii[ray];
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    const float gsize = gp.gsize[level];
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (uint32_t j = 0; j < cp.n; ++j) {
        const Gauss g = nlr_cast_one(cp, ray, k, j, t0, t1, o, d, bx, by, radius);
        const float werf = re_weights ? nlr_erf_weight(g.zs, gsize) : 1.0f;
        nlr_level_accum<T, C>(gp, level, g, werf, acc);
    }
    float *f = feat + (size_t)m * gp.L * C + level * C;
#pragma unroll
    for (int c = 0; c < C; ++c) f[c] = acc[c] / (float)cp.n;  // .mean(dim=-3), models.py:977
}

// ---------------------------------------------------------------------------------------------
// Proposal level: one thread per sample does every grid level and the tiny density MLP
// (L*C -> 64 -> 1, fp32 VALU with wave-uniform weights) and writes only density[M].
// ---------------------------------------------------------------------------------------------
struct PropMlpParams {
    const float *w1, *b1;  // dev [64, F], [64]
    const float *w2;       // dev [64]  (row 0 of density_layer.2)
    float b2, density_bias;
    uint32_t F;
};

template <typename T, int C, int LMAX>
__global__ void __launch_bounds__(256) nlr_prop_kernel(CastParams cp, GridParams gp, PropMlpParams mp, int re_weights,
                                                       float *__restrict__ density, float *__restrict__ feat_out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = cp.N * cp.S;
    if (m >= M) return;
    const uint32_t ray = m / cp.S, k = m - ray * cp.S;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float radius = cp.radii[ray];
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    float feat[LMAX * C];
#pragma unroll
    for (int i = 0; i < LMAX * C; ++i) feat[i] = 0.0f;
    for (uint32_t j = 0; j < cp.n; ++j) {
        const Gauss g = nlr_cast_one(cp, ray, k, j, t0, t1, o, d, bx, by, radius);
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < (int)gp.L) {
                const float werf = re_weights ? nlr_erf_weight(g.zs, gp.gsize[l]) : 1.0f;
                float a[C];
#pragma unroll
                for (int c = 0; c < C; ++c) a[c] = 0.0f;
                nlr_level_accum<T, C>(gp, l, g, werf, a);
#pragma unroll
                for (int c = 0; c < C; ++c) feat[l * C + c] += a[c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LMAX * C; ++i) feat[i] = feat[i] / (float)cp.n;
    if (feat_out)
        for (uint32_t i = 0; i < mp.F; ++i) feat_out[(size_t)m * mp.F + i] = feat[i];
    float raw = mp.b2;
    for (int h = 0; h < 64; ++h) {
        float a = mp.b1[h];
#pragma unroll
        for (int i = 0; i < LMAX * C; ++i)
            if (i < (int)mp.F) a = fmaf(mp.w1[h * mp.F + i], feat[i], a);
        raw = fmaf(mp.w2[h], fmaxf(a, 0.0f), raw);
    }
    const float x = raw + mp.density_bias;
    density[m] = x > 20.0f ? x : log1pf(expf(x));  // F.softplus (beta=1, threshold=20), models.py:1116
}

// Slot -> sample.  The fused kernels give 8 lanes (one per multisample) to a SLOT and 8 consecutive slots to a wave; which sample a slot
// is decides what a wave's 56 points have in common.
//   sample-major: slot = ray * S + k - a wave holds 8 consecutive samples of ONE ray.  Where the proposal chain piles a ray's samples up
//     (white-noise weights: every ray is absorbed within a few cells, at a depth unrelated to its neighbour's) they share their cells.
//   ray groups: inside a group of 8 consecutive rays the slots run ray-fastest, slot = (ray / 8) * 8 S + k * 8 + ray % 8 - a wave holds
//     8 ADJACENT rays at one sample index.  On a trained field the samples of a ray are metres apart (median interval 1.7 m) while
//     azimuth neighbours 0.18 m apart at 30 m meet the same surface at the same index: the table lines a wave gathers on the hashed
//     levels are shared.  Trained scene: encode 4.67 -> 3.37 ms, proposal 0.89 -> 0.60 ms; white noise: encode 1.33 -> 1.75 ms
//     (profiles/r04_ray_groups_ab.txt).
// Neither is right for every field, so a model render decides per level ON THE DEVICE (ray_groups = 2): nlr_ray_vote_kernel counts the
// (adjacent rays, sample index) pairs whose k-th samples lie within one interval of each other, and every workgroup of the encode launch
// reads the count.  Same arithmetic per sample either way: bit-identical features.  Rays beyond N (the last group of a ray count that is
// not a multiple of 8) are slots without a sample.
__device__ __forceinline__ bool nlr_slot_sample(const CastParams &cp, uint32_t slot, uint32_t &ray, uint32_t &k) {
    const bool groups = cp.ray_groups == 1u || (cp.ray_groups == 2u && cp.votes[0] > cp.vote_min);
    if (groups) {
        const uint32_t G = 8u * cp.S, gi = slot / G, w = slot - gi * G;
        ray = gi * 8u + (w & 7u);
        k = w >> 3;
    } else {
        ray = slot / cp.S;
        k = slot - ray * cp.S;
    }
    const bool in = ray < cp.N;
    if (!in) ray = cp.N - 1, k = cp.S - 1;   // (a harmless sample for the loads of a lane that contributes nothing)
    return in;
}
// (ray r, sample k), r < N - 1: does ray r + 1 put its k-th sample within ray r's k-th interval of it?  A fixed grid strides over the
// pairs and every workgroup adds its count once: atomics on ONE address queue at ~40 cycles each (profiles/r04_atomic_microbench.txt) -
// one per wave made this kernel 1.2 ms long.
__global__ void __launch_bounds__(256) nlr_ray_vote_kernel(const float *__restrict__ tdist, uint32_t N, uint32_t S, uint32_t *__restrict__ votes) {
    __shared__ uint32_t part[4];
    const uint32_t total = (N - 1) * S;
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t r = i / S, k = i - r * S;
        const float *a = tdist + (size_t)r * (S + 1) + k;
        mine += fabsf(a[S + 1] - a[0]) < (a[1] - a[0]) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c = part[0] + part[1] + part[2] + part[3];
        if (c) atomicAdd(votes, c);
    }
}
__global__ void nlr_ray_vote_zero_kernel(uint32_t *votes) { votes[threadIdx.x] = 0u; }
// tdist == NULL: zero the 64 counters of a render (before its first level)
int nlr_launch_ray_vote(const float *tdist, uint32_t N, uint32_t S, uint32_t *votes, hipStream_t st) {
    if (!tdist) {
        hipLaunchKernelGGL(nlr_ray_vote_zero_kernel, dim3(1), dim3(64), 0, st, votes);
        NLR_LAUNCH_CHECK("nlr_ray_vote_zero_kernel");
        return NLR_OK;
    }
    if (N < 2 || S == 0) return NLR_OK;
    NLR_CHECK_ARG((uint64_t)N * S < (1ull << 32), "ray vote: N * S = %llu does not fit 32 bits", (unsigned long long)N * S);
    const uint32_t nb = ((N - 1) * S + 255) / 256;
    hipLaunchKernelGGL(nlr_ray_vote_kernel, dim3(nb < 512u ? nb : 512u), dim3(256), 0, st, tdist, N, S, votes);
    NLR_LAUNCH_CHECK("nlr_ray_vote_kernel");
    return NLR_OK;
}
__host__ __forceinline__ uint64_t nlr_slot_count(const CastParams &cp) {
    return (uint64_t)(cp.ray_groups ? (cp.N + 7u) / 8u * 8u : cp.N) * cp.S;
}

// =============================================================================================================
// Multisample-parallel variants (sample_n <= 8): one lane per (sample, multisample), 8 lanes per sample.
// Cone casting + contraction run ONCE per multisample (the (sample, level) mapping above repeats them per level),
// every lane walks all grid levels with its 8*L gathers in flight, and the mean over the multisamples is a 3-step
// butterfly inside each 8-lane group.  7/8 of the lanes are active for sample_n = 7.
// =============================================================================================================
// Sum over the 8 lanes of a group, every lane gets it.  Three DPP adds (VALU) instead of three ds_bpermute round trips:
// quad_perm [1,0,3,2] and [2,3,0,1] are the xor-1 / xor-2 butterfly steps; after them the four lanes of a quad hold the
// same bits (a+b == b+a), so row_half_mirror (lane i <- lane 7-i of its 8-lane half row, i.e. some lane of the OTHER
// quad) delivers exactly what the xor-4 step would: the result is bit-identical to the __shfl_xor butterfly.
template <int CTRL>
__device__ __forceinline__ float nlr_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float nlr_group8_sum(float v) {
    v += nlr_dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += nlr_dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += nlr_dpp_mov<0x141>(v);  // row_half_mirror
    return v;
}

// Address of feature f = l * C + c of sample m.  piece_major (the fused path's internal layout, needs L * C % 4 == 0): [L*C/4][M][4] - a
// 4-float piece is C = 4: one level, C = 2: two levels, C = 1: four levels, C = 8: half a level; the 8 samples of a wave write one
// 128-byte run per piece and the MLP kernel's lanes (= samples) read 16 B each from consecutive addresses.  Otherwise row-major [M, L*C]
// (public nlr_mlp_level `features`).
template <int C>
__device__ __forceinline__ float *nlr_feat_ptr(float *feat, int piece_major, uint32_t M, uint32_t m, uint32_t l, uint32_t L) {
    if (piece_major) {
        const uint32_t f = l * C;
        return feat + ((size_t)(f >> 2) * M + m) * 4 + (f & 3u);
    }
    return feat + (size_t)m * L * C + l * C;
}

template <typename T, int C>
__global__ void __launch_bounds__(256) nlr_encode8g_kernel(CastParams cp, GridParams gp, int re_weights, float *__restrict__ feat,
                                                          int piece_major) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = cp.N * cp.S;
    const uint32_t j = gt & 7;
    uint32_t ray, k;
    const bool in = nlr_slot_sample(cp, gt >> 3, ray, k);
    const uint32_t m = ray * cp.S + k;
    const bool active = in && j < cp.n;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float radius = cp.radii[ray];
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    Gauss g = nlr_cast_one(cp, ray, k, active ? j : 0, t0, t1, o, d, bx, by, radius);
    if (!active) g.x0 = -1.0f;  // out of range -> contributes zeros
    const float inv_s8 = __frsqrt_rn(8.0f * (g.zs * g.zs));  // v_rsq_f32 (1 ulp) for 1/sqrtf (30 instructions)
    const float inv_n = 1.0f / (float)cp.n;
    for (uint32_t l = 0; l < gp.L; ++l) {
        const float werf = re_weights ? nlr_erf_weight_fast(inv_s8, gp.inv_gsize[l]) : 1.0f;
        float a[C];
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = 0.0f;
        nlr_level_accum<T, C>(gp, l, g, werf, a);
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = nlr_group8_sum(a[c]);
        if (in && j == (l & 7)) {  // spread the row stores over the lanes of the group
            // (layouts: nlr_feat_ptr)
            if constexpr (C == 8) {  // two pieces per level
                float *f0 = nlr_feat_ptr<4>(feat, piece_major, M, m, 2 * l, 2 * gp.L), *f1 = nlr_feat_ptr<4>(feat, piece_major, M, m, 2 * l + 1, 2 * gp.L);
#pragma unroll
                for (int c = 0; c < 4; ++c) f0[c] = a[c] * inv_n, f1[c] = a[4 + c] * inv_n;
            } else {
                float *f = nlr_feat_ptr<C>(feat, piece_major, M, m, l, gp.L);
#pragma unroll
                for (int c = 0; c < C; ++c) f[c] = a[c] * inv_n;
            }
        }
    }
}

// Proposal level: features as above, then the 64-unit density MLP split over the 8 lanes of the group (8 hidden
// units per lane) and a final butterfly for the raw density.
template <typename T, int C, int LMAX>
__global__ void __launch_bounds__(256) nlr_prop8g_kernel(CastParams cp, GridParams gp, PropMlpParams mp, int re_weights,
                                                        float *__restrict__ density, float *__restrict__ feat_out) {
    __shared__ float sw[64 * 16 + 128];  // w1 [64, F] | b1 [64] | w2 [64]
    for (uint32_t i = threadIdx.x; i < 64 * mp.F; i += 256) sw[i] = mp.w1[i];
    if (threadIdx.x < 64) {
        sw[64 * mp.F + threadIdx.x] = mp.b1[threadIdx.x];
        sw[64 * mp.F + 64 + threadIdx.x] = mp.w2[threadIdx.x];
    }
    __syncthreads();
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = gt & 7;
    uint32_t ray, k;
    const bool in = nlr_slot_sample(cp, gt >> 3, ray, k);
    const uint32_t m = ray * cp.S + k;
    const bool active = in && j < cp.n;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float radius = cp.radii[ray];
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    Gauss g = nlr_cast_one(cp, ray, k, active ? j : 0, t0, t1, o, d, bx, by, radius);
    if (!active) g.x0 = -1.0f;
    const float inv_s8 = __frsqrt_rn(8.0f * (g.zs * g.zs));  // v_rsq_f32 (1 ulp) for 1/sqrtf (30 instructions)
    const float inv_n = 1.0f / (float)cp.n;
    float feat[LMAX * C];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        float a[C];
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = 0.0f;
        if (l < (int)gp.L) {
            const float werf = re_weights ? nlr_erf_weight_fast(inv_s8, gp.inv_gsize[l]) : 1.0f;
            nlr_level_accum<T, C>(gp, l, g, werf, a);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) feat[l * C + c] = nlr_group8_sum(a[c]) * inv_n;  // every lane of the group gets the mean
    }
    if (feat_out && in && j == 0)
        for (uint32_t i = 0; i < mp.F; ++i) feat_out[(size_t)m * mp.F + i] = feat[i];
    // hidden unit hu = 8*hh + j: the 8 lanes of a group read 8 consecutive weight rows from the LDS copy (row stride
    // F floats -> at most 2-way bank conflicts), the 8 groups of a wave read the same addresses (broadcast)
    float part = 0.0f;
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
        const uint32_t hu = hh * 8 + j;
        float a = sw[64 * mp.F + hu];
#pragma unroll
        for (int i = 0; i < LMAX * C; ++i)
            if (i < (int)mp.F) a = fmaf(sw[hu * mp.F + i], feat[i], a);
        part = fmaf(sw[64 * mp.F + 64 + hu], fmaxf(a, 0.0f), part);
    }
    const float raw = nlr_group8_sum(part) + mp.b2;
    if (in && j == 0) {
        const float x = raw + mp.density_bias;
        density[m] = x > 20.0f ? x : log1pf(expf(x));  // F.softplus (beta=1, threshold=20), models.py:1116
    }
}

// =============================================================================================================
// Round-3 forms of the two kernels above on the level body of nlr_level_fast.h (grids it covers: linear interpolation,
// align_corners = False, every level dense or hashed with a power-of-two table - all grids of the path's configurations);
// the ...8g kernels stay for everything else.  Same lane mapping, same arithmetic per point; what changed is the
// instruction stream (see nlr_level_fast.h) and that a lane's validity (active multisample, point inside the unit cube) is decided
// once instead of once per level.
// =============================================================================================================
// Workgroup order.  The hardware deals the workgroups of a launch to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8) and every
// XCD has its own 4 MiB L2.  Consecutive workgroups here are consecutive 32-sample pieces of one ray and then of its neighbour in the batch
// (for a LiDAR sweep or an image: the next azimuth column / pixel), whose fine-level cells overlap; dealt round-robin, the lines they share
// are fetched into up to 8 L2s.  nlr_xcd_block hands every XCD CHUNKS of `chunk` consecutive logical workgroups instead: the b-th
// dispatched workgroup (the q-th of its XCD, q = b / 8) works on logical workgroup ((q / chunk) * 8 + xcd) * chunk + q mod chunk.  Chunks
// (not one eighth of the launch per XCD) keep the XCDs balanced when rays differ in cost.  The tail that does not fill 8 chunks keeps
// the identity order.
__device__ __forceinline__ uint32_t nlr_xcd_block(uint32_t b, uint32_t nblocks, uint32_t chunk) {
    const uint32_t per = 8u * chunk, full = (nblocks / per) * per;
    if (b >= full) return b;
    const uint32_t x = b & 7u, q = b >> 3;
    return ((q / chunk) * 8u + x) * chunk + (q % chunk);
}

#define NLR_XCD_CHUNK 32u  // workgroups (of 32 samples) per chunk: 8 rays of a 128-sample level (profiles/r04_encode_experiments.txt)

#ifdef NLR_DBG_ENV
// Diagnostic builds only (scripts/diag_encode.sh; never in libnerflidar_hip.so): experiment switches of the gather kernels, set from the
// environment at launch (NLR_ENC_LO / _HI / _NOSCALAR / _NT; NLR_ENC_CHUNK and NLR_ENC_PERSIST are read by the launchers).
// [0] first level, [1] one past the last level, [3] scalar path off, [4] bit mask of levels gathered non-temporally
// (the array itself is defined in nlr_level_fast.h, which reads [3])
#define NLR_DBG_LO nlr_dbg[0]
#define NLR_DBG_HI nlr_dbg[1]
// experiment: processing order in 2-D tiles of the sweep ([2] = azimuth columns W, [5] x [6] = beams x columns per tile): the po-th ray
// processed is ray beam * W + az of the tile walk
__device__ __forceinline__ uint32_t nlr_dbg_ray(uint32_t po) {
    const uint32_t W = (uint32_t)nlr_dbg[2], tb = (uint32_t)nlr_dbg[5], ta = (uint32_t)nlr_dbg[6];
    if (!W || !tb || !ta) return po;
    const uint32_t per = tb * ta, tile = po / per, within = po - tile * per, tiles_row = W / ta;
    const uint32_t trow = tile / tiles_row, tcol = tile - trow * tiles_row;
    return (trow * tb + within / ta) * W + tcol * ta + within % ta;
}
#define NLR_DBG_RAY(r) nlr_dbg_ray(r)
#else
#define NLR_DBG_LO 0
#define NLR_DBG_HI 99
#define NLR_DBG_RAY(r) (r)
#endif

struct RayRegs {
    float o[3], d[3], bx[3], by[3], radius, t0, t1;
};
__device__ __forceinline__ RayRegs nlr_load_ray(const CastParams &cp, uint32_t ray, uint32_t k) {
    RayRegs r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r.o[c] = cp.origins[(size_t)ray * 3 + c];
        r.d[c] = cp.directions[(size_t)ray * 3 + c];
        r.bx[c] = cp.base_x[(size_t)ray * 3 + c];
        r.by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    r.radius = cp.radii[ray];
    r.t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k];
    r.t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    return r;
}

template <typename T, int C>
__device__ __forceinline__ void nlr_encode8_block(const CastParams &cp, const GridParams &gp, int re_weights, float *__restrict__ feat,
                                                  int piece_major, uint32_t block) {
    const uint32_t gt = block * 256u + threadIdx.x;
    const uint32_t M = cp.N * cp.S;
    const uint32_t j = gt & 7;
    uint32_t pray, k;
    const bool in = nlr_slot_sample(cp, gt >> 3, pray, k);
    const uint32_t ray = NLR_DBG_RAY(pray);
    const uint32_t m = ray * cp.S + k;
    const bool active = in && j < cp.n;
    const RayRegs rr = nlr_load_ray(cp, ray, k);
    const Gauss g = nlr_cast_one(cp, ray, k, active ? j : 0, rr.t0, rr.t1, rr.o, rr.d, rr.bx, rr.by, rr.radius);
    // gridencoder.cu:124-135: a point outside [0,1]^3 encodes as zeros on every level
    const bool valid = active && !((g.x0 < 0 || g.x0 > 1) || (g.x1 < 0 || g.x1 > 1) || (g.x2 < 0 || g.x2 > 1));
    const float inv_s8 = __frsqrt_rn(8.0f * (g.zs * g.zs));  // v_rsq_f32 (1 ulp) for 1/sqrtf (30 instructions)
    const float inv_n = 1.0f / (float)cp.n;
    float r[C];  // this lane's contribution to the level; stays 0 in lanes without a point
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0.0f;
    for (uint32_t l = NLR_DBG_LO; l < gp.L && l < (uint32_t)NLR_DBG_HI; ++l) {
        if (valid) {
            const float werf = re_weights ? nlr_erf_weight_fast(inv_s8, gp.inv_gsize[l]) : 1.0f;
            if (gp.mode[l] == 0) nlr_level_fast<T, C, 0>(gp, l, g.x0, g.x1, g.x2, werf, r);
            else nlr_level_fast<T, C, 1>(gp, l, g.x0, g.x1, g.x2, werf, r);
        }
        float a[C];
        nlr_group8_sum_to(r, a);
        if (in && j == (l & 7)) {  // spread the row stores over the lanes of the group (layouts: see nlr_encode8g_kernel)
            if constexpr (C == 8) {  // two pieces per level
                float *f0 = nlr_feat_ptr<4>(feat, piece_major, M, m, 2 * l, 2 * gp.L), *f1 = nlr_feat_ptr<4>(feat, piece_major, M, m, 2 * l + 1, 2 * gp.L);
#pragma unroll
                for (int c = 0; c < 4; ++c) f0[c] = a[c] * inv_n, f1[c] = a[4 + c] * inv_n;
                continue;
            }
            float *f = nlr_feat_ptr<C>(feat, piece_major, M, m, l, gp.L);
            if constexpr (C == 4) {
                // one 16-byte non-temporal store: the features are a 0.67 GB stream per sweep that the MLP kernel reads once; kept out of
                // the L2's recently-used set they do not evict table lines the next samples will gather again
                const nlr_f4 q = {a[0] * inv_n, a[1] * inv_n, a[2] * inv_n, a[3] * inv_n};
                __builtin_nontemporal_store(q, (nlr_f4 *)f);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) f[c] = a[c] * inv_n;
            }
        }
    }
}

// One workgroup per 32 samples, or (nblocks > gridDim.x) a persistent grid that walks the logical workgroups with the grid's stride; the
// dispatch slot decides the XCD, nlr_xcd_block the samples (see there).
template <typename T, int C>
__global__ void __launch_bounds__(256) nlr_encode8_kernel(CastParams cp, GridParams gp, int re_weights, float *__restrict__ feat,
                                                          int piece_major, uint32_t nblocks, uint32_t chunk) {
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x)
        nlr_encode8_block<T, C>(cp, gp, re_weights, feat, piece_major, chunk ? nlr_xcd_block(b, nblocks, chunk) : b);
}

template <typename T, int C, int LMAX>
__global__ void __launch_bounds__(256) nlr_prop8_kernel(CastParams cp, GridParams gp, PropMlpParams mp, int re_weights,
                                                        float *__restrict__ density, float *__restrict__ feat_out, uint32_t nblocks, uint32_t chunk) {
    __shared__ float sw[64 * 16 + 128];  // w1 [64, F] | b1 [64] | w2 [64]
    for (uint32_t i = threadIdx.x; i < 64 * mp.F; i += 256) sw[i] = mp.w1[i];
    if (threadIdx.x < 64) {
        sw[64 * mp.F + threadIdx.x] = mp.b1[threadIdx.x];
        sw[64 * mp.F + 64 + threadIdx.x] = mp.w2[threadIdx.x];
    }
    __syncthreads();
  for (uint32_t lb = blockIdx.x; lb < nblocks; lb += gridDim.x) {  // (one workgroup per 32 samples, or a persistent grid: see nlr_encode8_kernel)
    const uint32_t gt = (chunk ? nlr_xcd_block(lb, nblocks, chunk) : lb) * 256u + threadIdx.x;
    const uint32_t j = gt & 7;
    uint32_t pray, k;
    const bool in = nlr_slot_sample(cp, gt >> 3, pray, k);
    const uint32_t ray = NLR_DBG_RAY(pray);
    const uint32_t m = ray * cp.S + k;
    const bool active = in && j < cp.n;
    const RayRegs rr = nlr_load_ray(cp, ray, k);
    const Gauss g = nlr_cast_one(cp, ray, k, active ? j : 0, rr.t0, rr.t1, rr.o, rr.d, rr.bx, rr.by, rr.radius);
    const bool valid = active && !((g.x0 < 0 || g.x0 > 1) || (g.x1 < 0 || g.x1 > 1) || (g.x2 < 0 || g.x2 > 1));
    const float inv_s8 = __frsqrt_rn(8.0f * (g.zs * g.zs));
    const float inv_n = 1.0f / (float)cp.n;
    float feat[LMAX * C];
#pragma unroll
    for (int i = 0; i < LMAX * C; ++i) feat[i] = 0.0f;
    if (valid) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < (int)gp.L) {
                const float werf = re_weights ? nlr_erf_weight_fast(inv_s8, gp.inv_gsize[l]) : 1.0f;
                float a[C];
                if (gp.mode[l] == 0) nlr_level_fast<T, C, 0>(gp, l, g.x0, g.x1, g.x2, werf, a);
                else nlr_level_fast<T, C, 1>(gp, l, g.x0, g.x1, g.x2, werf, a);
#pragma unroll
                for (int c = 0; c < C; ++c) feat[l * C + c] = a[c];
            }
        }
    }
    // multisample means of all levels, four values per butterfly block; every lane of the group gets them
    static_assert((LMAX * C) % 4 == 0, "feature count of the proposal kernel must be a multiple of 4");
#pragma unroll
    for (int i = 0; i < LMAX * C; i += 4) {
        float q[4] = {feat[i], feat[i + 1], feat[i + 2], feat[i + 3]};
        nlr_group8_sum_n(q);
#pragma unroll
        for (int u = 0; u < 4; ++u) feat[i + u] = q[u] * inv_n;
    }
    if (feat_out && in && j == 0)
        for (uint32_t i = 0; i < mp.F; ++i) feat_out[(size_t)m * mp.F + i] = feat[i];
    // hidden unit hu = 8*hh + j: see nlr_prop8g_kernel
    float part = 0.0f;
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
        const uint32_t hu = hh * 8 + j;
        float a = sw[64 * mp.F + hu];
#pragma unroll
        for (int i = 0; i < LMAX * C; ++i)
            if (i < (int)mp.F) a = fmaf(sw[hu * mp.F + i], feat[i], a);
        part = fmaf(sw[64 * mp.F + 64 + hu], fmaxf(a, 0.0f), part);
    }
    float praw[1] = {part};
    nlr_group8_sum_n(praw);
    if (in && j == 0) {
        const float x = (praw[0] + mp.b2) + mp.density_bias;
        density[m] = x > 20.0f ? x : log1pf(expf(x));  // F.softplus (beta=1, threshold=20), models.py:1116
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of the fused features (training path), first half: per multisample point, its unit-cube position and the gradient of
// its level features, d_feat[m, l, :] * w_erf_j(l) / n - cast + contraction + re-weighting recomputed, one lane per (sample,
// multisample).  The scatter into the table is then nlr_grid_encode_backward (run-length aggregated atomics, LDS accumulation of
// the small dense levels).  A scatter fused into this kernel (one lane walking all levels, or blockIdx.y = level) was built and
// measured 3-4x slower than the two-kernel form: it loses the LDS accumulation of the hot dense levels and half of the run
// lengths (an inactive 8th lane in every group of multisamples).
// ---------------------------------------------------------------------------------------------
int nlr_fill_cast_params(CastParams *cp, const NlrRays *rays, const float *tdist, const float *rand_deg, uint32_t N,
                         uint32_t S, uint32_t n, uint32_t mloops, float std_scale);
int nlr_launch_encode(const CastParams &cp, const GridParams &gp, int re_weights, float *feat, int piece_major, hipStream_t st);

__global__ void __launch_bounds__(256) nlr_encode_expand_kernel(CastParams cp, GridParams gp, int re_weights, const float *__restrict__ d_feat,
                                                               float *__restrict__ x01, float *__restrict__ g_pts) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = (size_t)cp.N * cp.S;
    if (t >= M * cp.n) return;
    const uint32_t m = (uint32_t)(t / cp.n), j = (uint32_t)(t - (size_t)m * cp.n);
    const uint32_t ray = m / cp.S, k = m - ray * cp.S;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    const Gauss g = nlr_cast_one(cp, ray, k, j, t0, t1, o, d, bx, by, cp.radii[ray]);
    x01[t * 3 + 0] = g.x0;
    x01[t * 3 + 1] = g.x1;
    x01[t * 3 + 2] = g.x2;
    const float inv_s8 = __frsqrt_rn(8.0f * (g.zs * g.zs));
    const float inv_n = 1.0f / (float)cp.n;
    const uint32_t LC = gp.L * gp.C;
    const float *gf = d_feat + (size_t)m * LC;
    float *out = g_pts + t * LC;
    for (uint32_t l = 0; l < gp.L; ++l) {
        const float w = (re_weights ? nlr_erf_weight_fast(inv_s8, gp.inv_gsize[l]) : 1.0f) * inv_n;
        for (uint32_t c = 0; c < gp.C; ++c) out[l * gp.C + c] = gf[l * gp.C + c] * w;
    }
}

static int encode_features_args(CastParams *cp, GridParams *gp, const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n,
                                uint32_t sample_m, float std_scale, const float *rand_deg, const NlrGridDesc *grid) {
    NLR_CHECK_ARG(grid && grid->table && grid->offsets && tdist, "encode_features: NULL argument");
    int rc = nlr_fill_cast_params(cp, rays, tdist, rand_deg, N, S, sample_n, sample_m, std_scale);
    if (rc) return rc;
    return nlr_fill_grid_params(gp, grid->table, grid->table_dtype, grid->offsets, grid->num_levels, grid->level_dim, grid->log2_per_level_scale,
                                grid->base_resolution, grid->gridtype, (int)grid->align_corners, grid->interp);
}

extern "C" int nlr_encode_features_forward(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                           float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights, float *features,
                                           void *stream) {
    if (N == 0 || S == 0) return NLR_OK;
    NLR_CHECK_ARG(features, "encode_features_forward: NULL output");
    CastParams cp;
    GridParams gp;
    int rc = encode_features_args(&cp, &gp, rays, tdist, N, S, sample_n, sample_m, std_scale, rand_deg, grid);
    if (rc) return rc;
    return nlr_launch_encode(cp, gp, (int)re_weights, features, 0, (hipStream_t)stream);
}

extern "C" int nlr_encode_features_backward_ws(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                               float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights,
                                               const float *d_features, float *points_tmp, float *grad_tmp, float *grad_table, void *workspace,
                                               size_t workspace_bytes, void *stream);

extern "C" int nlr_encode_features_backward(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                            float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights,
                                            const float *d_features, float *points_tmp, float *grad_tmp, float *grad_table, void *stream) {
    return nlr_encode_features_backward_ws(rays, tdist, N, S, sample_n, sample_m, std_scale, rand_deg, grid, re_weights, d_features, points_tmp,
                                           grad_tmp, grad_table, nullptr, 0, stream);
}

extern "C" int nlr_encode_features_backward_ws(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n, uint32_t sample_m,
                                               float std_scale, const float *rand_deg, const NlrGridDesc *grid, uint32_t re_weights,
                                               const float *d_features, float *points_tmp, float *grad_tmp, float *grad_table, void *workspace,
                                               size_t workspace_bytes, void *stream) {
    if (N == 0 || S == 0) return NLR_OK;
    NLR_CHECK_ARG(d_features && grad_table && points_tmp && grad_tmp, "encode_features_backward: NULL tensor");
    CastParams cp;
    GridParams gp;
    int rc = encode_features_args(&cp, &gp, rays, tdist, N, S, sample_n, sample_m, std_scale, rand_deg, grid);
    if (rc) return rc;
    const size_t B = (size_t)N * S * sample_n;
    NLR_CHECK_ARG(B < (1ull << 32), "encode_features_backward: %zu multisample points do not fit the 32-bit point index", B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nlr_encode_expand_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, cp, gp, (int)re_weights, d_features, points_tmp,
                       grad_tmp);
    NLR_LAUNCH_CHECK("nlr_encode_expand_kernel");
    return nlr_grid_encode_backward_ws(grad_tmp, points_tmp, grid->offsets, grad_table, (uint32_t)B, 3, grid->level_dim, grid->num_levels,
                                       grid->log2_per_level_scale, grid->base_resolution, nullptr, nullptr, grid->gridtype, (int)grid->align_corners,
                                       grid->interp, 1, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// Rows a-5 / a-6 alone (training path, nerflidar_hip/training.py): the contracted multisample means / bound and stds / bound
// that MLP.predict_density hands to the GridEncoder (ZI/models.py:965-973), one thread per (sample, multisample).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nlr_cast_contract_kernel(CastParams cp, float *__restrict__ means, float *__restrict__ stds) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = (size_t)cp.N * cp.S;
    if (t >= M * cp.n) return;
    const uint32_t m = (uint32_t)(t / cp.n), j = (uint32_t)(t - (size_t)m * cp.n);
    const uint32_t ray = m / cp.S, k = m - ray * cp.S;
    float o[3], d[3], bx[3], by[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cp.origins[(size_t)ray * 3 + c];
        d[c] = cp.directions[(size_t)ray * 3 + c];
        bx[c] = cp.base_x[(size_t)ray * 3 + c];
        by[c] = cp.base_y[(size_t)ray * 3 + c];
    }
    const float t0 = cp.tdist[(size_t)ray * (cp.S + 1) + k], t1 = cp.tdist[(size_t)ray * (cp.S + 1) + k + 1];
    float raw[3];
    const Gauss g = nlr_cast_one(cp, ray, k, j, t0, t1, o, d, bx, by, cp.radii[ray], raw);
    means[t * 3 + 0] = raw[0];
    means[t * 3 + 1] = raw[1];
    means[t * 3 + 2] = raw[2];
    stds[t] = g.zs;
}

int nlr_fill_cast_params(CastParams *cp, const NlrRays *rays, const float *tdist, const float *rand_deg, uint32_t N,
                         uint32_t S, uint32_t n, uint32_t mloops, float std_scale);

extern "C" int nlr_cast_contract(const NlrRays *rays, const float *tdist, uint32_t N, uint32_t S, uint32_t sample_n,
                                 uint32_t sample_m, float std_scale, const float *rand_deg, float *means, float *stds, void *stream) {
    NLR_CHECK_ARG(tdist && means && stds, "cast_contract: NULL tensor");
    if (N == 0 || S == 0) return NLR_OK;
    CastParams cp;
    int rc = nlr_fill_cast_params(&cp, rays, tdist, rand_deg, N, S, sample_n, sample_m, std_scale);
    if (rc) return rc;
    const size_t total = (size_t)N * S * sample_n;
    hipLaunchKernelGGL(nlr_cast_contract_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cp, means, stds);
    NLR_LAUNCH_CHECK("nlr_cast_contract_kernel");
    return NLR_OK;
}

// ---- host side ---------------------------------------------------------------------------------
int nlr_fill_cast_params(CastParams *cp, const NlrRays *rays, const float *tdist, const float *rand_deg, uint32_t N,
                         uint32_t S, uint32_t n, uint32_t mloops, float std_scale) {
    NLR_CHECK_ARG(n >= 1 && n <= NLR_MAX_MULTI, "cast: sample_n=%u outside [1,%d]", n, NLR_MAX_MULTI);
    NLR_CHECK_ARG(rays && rays->origins && rays->directions && rays->radii && rays->base_x && rays->base_y,
                  "cast: ray batch has NULL origins/directions/radii/base_x/base_y");
    memset(cp, 0, sizeof(*cp));
    cp->origins = rays->origins;
    cp->directions = rays->directions;
    cp->base_x = rays->base_x;
    cp->base_y = rays->base_y;
    cp->radii = rays->radii;
    cp->tdist = tdist;
    cp->rand_deg = rand_deg;
    cp->N = N;
    cp->S = S;
    cp->n = n;
    cp->std_scale = std_scale;
    for (uint32_t j = 0; j < n; ++j) {
        // `2 * torch.pi * m * j / n` with j an int64 tensor: float32(2*pi*m) * float32(j), then / n
        const float deg = ((float)(2.0 * M_PI * (double)mloops) * (float)j) / (float)n;
        cp->degj[j] = deg;
        cp->cosd[j] = cosf(deg);
        cp->sind[j] = sinf(deg);
    }
    return NLR_OK;
}

// Diagnostic switches (include/nerflidar_hip.h: nlr_debug_set / nlr_debug_get): explicit and readable back; the library reads no
// environment variable.  [0] force the generic level body, [1] cap of the persistent MLP grid for models created afterwards.
#include <atomic>
static std::atomic<int> nlr_debug_switch[7] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}};
extern "C" int nlr_debug_set(uint32_t key, int value) {
    NLR_CHECK_ARG(key < 7, "debug_set: unknown key %u", key);
    nlr_debug_switch[key].store(value);
    return NLR_OK;
}
extern "C" int nlr_debug_get(uint32_t key) { return key < 7 ? nlr_debug_switch[key].load() : 0; }
static bool nlr_force_generic() { return nlr_debug_switch[NLR_DBG_FORCE_GENERIC].load() != 0; }

extern "C" int nlr_grid_fast_path(const int32_t *offsets_host, uint32_t L, uint32_t C, float S, uint32_t H, int table_dtype, uint32_t gridtype,
                                  int align_corners, uint32_t interp) {
    GridParams gp;
    if (!offsets_host || nlr_fill_grid_params(&gp, offsets_host, table_dtype, offsets_host, L, C, S, H, gridtype, align_corners, interp)) return 0;
    return nlr_level_fast_ok(gp) ? 1 : 0;
}

#ifdef NLR_DBG_ENV
static void nlr_dbg_upload(hipStream_t st) {
    int v[8] = {0, 99, 0, 0, 0, 0, 0, 0};
    if (const char *e = getenv("NLR_ENC_LO")) v[0] = atoi(e);
    if (const char *e = getenv("NLR_ENC_HI")) v[1] = atoi(e);
    if (const char *e = getenv("NLR_ENC_NOSCALAR")) v[3] = atoi(e);
    if (const char *e = getenv("NLR_ENC_NT")) v[4] = (int)strtol(e, nullptr, 0);
    if (const char *e = getenv("NLR_ENC_SWEEPW")) v[2] = atoi(e);
    if (const char *e = getenv("NLR_ENC_TILEB")) v[5] = atoi(e);
    if (const char *e = getenv("NLR_ENC_TILEA")) v[6] = atoi(e);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(nlr_dbg), v, sizeof(v), 0, hipMemcpyHostToDevice, st);
}
#endif

int nlr_launch_encode(const CastParams &cp, const GridParams &gp, int re_weights, float *feat, int piece_major, hipStream_t st) {
#ifdef NLR_DBG_ENV
    nlr_dbg_upload(st);
#endif
    // 8 lanes per sample with a 32-bit lane index (the kernels compute N * S in 32 bits)
    NLR_CHECK_ARG(nlr_slot_count(cp) * 8 < (1ull << 32), "encode: N * S = %llu samples do not fit the 32-bit lane index (chunk the rays)",
                  (unsigned long long)cp.N * cp.S);
    const uint32_t M = cp.N * cp.S;
    if (cp.n <= 8) {  // multisample-parallel mapping
        dim3 grid8((uint32_t)((nlr_slot_count(cp) * 8 + 255) / 256)), block8(256);
        const bool fast = nlr_level_fast_ok(gp) && !nlr_force_generic();
        uint32_t chunk = NLR_XCD_CHUNK;
        dim3 gridp = grid8;
#ifdef NLR_DBG_ENV
        if (const char *e = getenv("NLR_ENC_CHUNK")) chunk = (uint32_t)atoi(e);
        if (const char *e = getenv("NLR_ENC_PERSIST")) gridp.x = (uint32_t)atoi(e) * 256u < grid8.x ? (uint32_t)atoi(e) * 256u : grid8.x;
#endif
#define NLR_ENC8(T, C)                                                                                                               \
    do {                                                                                                                             \
        if (fast) hipLaunchKernelGGL((nlr_encode8_kernel<T, C>), gridp, block8, 0, st, cp, gp, re_weights, feat, piece_major, grid8.x, chunk); \
        else hipLaunchKernelGGL((nlr_encode8g_kernel<T, C>), grid8, block8, 0, st, cp, gp, re_weights, feat, piece_major);           \
    } while (0)
        if (gp.table_dtype == 0) {
            switch (gp.C) {
                case 1: NLR_ENC8(float, 1); break;
                case 2: NLR_ENC8(float, 2); break;
                case 4: NLR_ENC8(float, 4); break;
                default: NLR_ENC8(float, 8); break;
            }
        } else {
            switch (gp.C) {
                case 1: NLR_ENC8(__half, 1); break;
                case 2: NLR_ENC8(__half, 2); break;
                case 4: NLR_ENC8(__half, 4); break;
                default: NLR_ENC8(__half, 8); break;
            }
        }
#undef NLR_ENC8
        NLR_LAUNCH_CHECK("nlr_encode8_kernel");
        return NLR_OK;
    }
    dim3 grid((M + 255) / 256, gp.L), block(256);
#define NLR_ENC(T, C) hipLaunchKernelGGL((nlr_encode_kernel<T, C>), grid, block, 0, st, cp, gp, re_weights, feat)
    if (gp.table_dtype == 0) {
        switch (gp.C) {
            case 1: NLR_ENC(float, 1); break;
            case 2: NLR_ENC(float, 2); break;
            case 4: NLR_ENC(float, 4); break;
            default: NLR_ENC(float, 8); break;
        }
    } else {
        switch (gp.C) {
            case 1: NLR_ENC(__half, 1); break;
            case 2: NLR_ENC(__half, 2); break;
            case 4: NLR_ENC(__half, 4); break;
            default: NLR_ENC(__half, 8); break;
        }
    }
#undef NLR_ENC
    NLR_LAUNCH_CHECK("nlr_encode_kernel");
    return NLR_OK;
}

int nlr_launch_prop(const CastParams &cp, const GridParams &gp, const float *w1, const float *b1, const float *w2, float b2,
                    float density_bias, int re_weights, float *density, float *feat_out, hipStream_t st) {
    NLR_CHECK_ARG(nlr_slot_count(cp) * 8 < (1ull << 32), "proposal level: N * S = %llu samples do not fit the 32-bit lane index (chunk the rays)",
                  (unsigned long long)cp.N * cp.S);
    const uint32_t M = cp.N * cp.S;
#ifdef NLR_DBG_ENV
    nlr_dbg_upload(st);
#endif
    PropMlpParams mp;
    mp.w1 = w1;
    mp.b1 = b1;
    mp.w2 = w2;
    mp.b2 = b2;
    mp.density_bias = density_bias;
    mp.F = gp.L * gp.C;
    NLR_CHECK_ARG(mp.F <= 16, "proposal MLP: L*C = %u > 16 features is outside the fused proposal kernel", mp.F);
    if (cp.n <= 8) {
        dim3 grid8((uint32_t)((nlr_slot_count(cp) * 8 + 255) / 256)), block8(256);
        const bool fast = nlr_level_fast_ok(gp) && !nlr_force_generic();
        uint32_t chunk = NLR_XCD_CHUNK;
        dim3 gridp = grid8;
#ifdef NLR_DBG_ENV
        if (const char *e = getenv("NLR_ENC_CHUNK")) chunk = (uint32_t)atoi(e);
        if (const char *e = getenv("NLR_ENC_PERSIST")) gridp.x = (uint32_t)atoi(e) * 256u < grid8.x ? (uint32_t)atoi(e) * 256u : grid8.x;
#endif
#define NLR_PROP8(T, C, LM)                                                                                                                    \
    do {                                                                                                                                       \
        if (fast) hipLaunchKernelGGL((nlr_prop8_kernel<T, C, LM>), gridp, block8, 0, st, cp, gp, mp, re_weights, density, feat_out, grid8.x, chunk); \
        else hipLaunchKernelGGL((nlr_prop8g_kernel<T, C, LM>), grid8, block8, 0, st, cp, gp, mp, re_weights, density, feat_out);               \
    } while (0)
        const bool f32t = gp.table_dtype == 0;
        if (gp.C == 1 && gp.L <= 8) { if (f32t) NLR_PROP8(float, 1, 8); else NLR_PROP8(__half, 1, 8); }
        else if (gp.C == 1) { if (f32t) NLR_PROP8(float, 1, 16); else NLR_PROP8(__half, 1, 16); }
        else if (gp.C == 2) { if (f32t) NLR_PROP8(float, 2, 8); else NLR_PROP8(__half, 2, 8); }
        else if (gp.C == 4 && gp.L <= 4) { if (f32t) NLR_PROP8(float, 4, 4); else NLR_PROP8(__half, 4, 4); }
        else NLR_FAIL(NLR_ERR_UNSUPPORTED, "proposal MLP: grid L=%u C=%u not supported by the fused proposal kernel", gp.L, gp.C);
#undef NLR_PROP8
        NLR_LAUNCH_CHECK("nlr_prop8_kernel");
        return NLR_OK;
    }
    dim3 grid((M + 255) / 256), block(256);
#define NLR_PROP(T, C, LM) hipLaunchKernelGGL((nlr_prop_kernel<T, C, LM>), grid, block, 0, st, cp, gp, mp, re_weights, density, feat_out)
    const bool f32 = gp.table_dtype == 0;
    if (gp.C == 1 && gp.L <= 8) { if (f32) NLR_PROP(float, 1, 8); else NLR_PROP(__half, 1, 8); }
    else if (gp.C == 1) { if (f32) NLR_PROP(float, 1, 16); else NLR_PROP(__half, 1, 16); }
    else if (gp.C == 2) { if (f32) NLR_PROP(float, 2, 8); else NLR_PROP(__half, 2, 8); }
    else if (gp.C == 4 && gp.L <= 4) { if (f32) NLR_PROP(float, 4, 4); else NLR_PROP(__half, 4, 4); }
    else NLR_FAIL(NLR_ERR_UNSUPPORTED, "proposal MLP: grid L=%u C=%u not supported by the fused proposal kernel", gp.L, gp.C);
#undef NLR_PROP
    NLR_LAUNCH_CHECK("nlr_prop_kernel");
    return NLR_OK;
}
