// One multiresolution-grid level for one point: the 8-corner trilinear gather of gridencoder.cu:137-197, shared by the fused
// cast + encode kernels (nlr_encode.hip) and the object networks (nlr_objects.hip).
#pragma once
#include "nlr_kernels.h"

#include <hip/hip_fp16.h>

struct Gauss {  // one contracted multisample, mapped to the unit cube
    float x0, x1, x2, zs;
};

template <typename T>
__device__ __forceinline__ float nlr_ld(const T *p);
template <>
__device__ __forceinline__ float nlr_ld<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float nlr_ld<__half>(const __half *p) { return __half2float(*p); }

// Trilinear interpolation of C channels at one level (gridencoder.cu:137-197); acc += w_erf * value.
template <typename T, int C, int MODE>
__device__ __forceinline__ void nlr_level_accum_m(const GridParams &gp, uint32_t level, const Gauss &g, float werf, float (&acc)[C]) {
    if ((g.x0 < 0 || g.x0 > 1) || (g.x1 < 0 || g.x1 > 1) || (g.x2 < 0 || g.x2 > 1)) return;  // zeros
    const T *grid = (const T *)gp.table + (size_t)gp.offset[level] * C;
    const float scale = gp.scale[level];
    const float half = gp.align_corners ? 0.0f : 0.5f;
    float pos[3] = {fmaf(g.x0, scale, half), fmaf(g.x1, scale, half), fmaf(g.x2, scale, half)};
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (gp.interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    }
    uint32_t idx[8];
    nlr_corner_idx<MODE>(gp, level, pg, idx);
    float v[8][C];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
        for (int c = 0; c < C; ++c) v[c8][c] = nlr_ld<T>(grid + (size_t)idx[c8] * C + c);
    // corner weights in the reference's multiplication order ((wx * wy) * wz)
    const float wx[2] = {1 - pos[0], pos[0]}, wy[2] = {1 - pos[1], pos[1]}, wz[2] = {1 - pos[2], pos[2]};
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0.0f;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const float w = (wx[c8 & 1] * wy[(c8 >> 1) & 1]) * wz[(c8 >> 2) & 1];
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] = fmaf(w, v[c8][c], r[c]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += r[c] * werf;
}

// `mode` is wave-uniform (a property of the level)
template <typename T, int C>
__device__ __forceinline__ void nlr_level_accum(const GridParams &gp, uint32_t level, const Gauss &g, float werf, float (&acc)[C]) {
    const uint32_t mode = gp.mode[level];
    if (mode == 0) nlr_level_accum_m<T, C, 0>(gp, level, g, werf, acc);
    else if (mode == 1) nlr_level_accum_m<T, C, 1>(gp, level, g, werf, acc);
    else nlr_level_accum_m<T, C, 2>(gp, level, g, werf, acc);
}


// ---- atomics of a backward pass over ray-ordered points -----------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float nlr_dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Run-length aggregation.  The points of a batch arrive in ray order (ray, sample, multisample): neighbouring lanes of a channel
// are neighbouring points on one ray and, on every level coarser than their spacing, fall into the same cell - a wave would send
// runs of atomics to one address, which the memory pipeline serialises.  A segmented inclusive scan over the runs of equal
// address inside each 16-lane row (DPP row_shr; the kernel lays a wave out channel-major, so the stride is 1 and lanes of
// different channels never share an address) leaves every run's sum in its last lane, and only that lane issues the atomic.  Correct for any address sequence: only contiguous equal addresses are merged.
template <int CTRL>
__device__ __forceinline__ uint32_t nlr_dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int C>
__device__ __forceinline__ void nlr_run_atomic(float *gt, uint32_t addr, float v, bool valid, int lane) {
    const int r = lane & 15;
    const uint32_t key = valid ? addr : 0xffffffffu - (uint32_t)lane;  // invalid lanes never match a neighbour
    if (!valid) v = 0.0f;
    // head of a run: no lane C to the left in the row, or a different address there
    constexpr int SHR = 0x110;  // row_shr:n = 0x110 + n
    constexpr int SHL = 0x100;  // row_shl:n = 0x100 + n
    const uint32_t left = nlr_dpp_u<SHR + C>(key);
    uint32_t head = (r < C || left != key) ? 1u : 0u;
    const uint32_t right_head = nlr_dpp_u<SHL + C>(head);
    const bool tail = (r >= 16 - C) || right_head != 0u;
#define NLR_RUN_STEP(D)                                          \
    if constexpr ((D) < 16) {                                      \
        const float vo = nlr_dpp_f<SHR + (D)>(v);                 \
        const uint32_t hs = nlr_dpp_u<SHR + (D)>(head);           \
        const uint32_t ho = r < (D) ? 1u : hs;                    \
        if (!head) {                                               \
            v += vo;                                               \
            head = ho;                                             \
        }                                                          \
    }
    NLR_RUN_STEP(C)
    NLR_RUN_STEP(2 * C)
    NLR_RUN_STEP(4 * C)
    NLR_RUN_STEP(8 * C)
#undef NLR_RUN_STEP
    if (valid && tail) atomicAdd(gt + addr, v);
}

// Segmented inclusive scan over runs of equal `key` inside a 16-lane row, C values per lane; true in the lane that ends a run
// (it then holds the run's sums).  Lanes with valid = false never join a run.
template <int C>
__device__ __forceinline__ bool nlr_run_merge(uint32_t key, float (&v)[C], bool valid, int lane) {
    const int r = lane & 15;
    const uint32_t k = valid ? key : 0xffffffffu - (uint32_t)lane;
    constexpr int SHR = 0x110, SHL = 0x100;
    const uint32_t left = nlr_dpp_u<SHR + 1>(k);
    uint32_t head = (r < 1 || left != k) ? 1u : 0u;
    const uint32_t right_head = nlr_dpp_u<SHL + 1>(head);
    const bool tail = (r >= 15) || right_head != 0u;
#define NLR_MERGE_STEP(D)                                              \
    {                                                                  \
        float vo[C];                                                   \
        _Pragma("unroll") for (int c = 0; c < C; ++c) vo[c] = nlr_dpp_f<SHR + (D)>(v[c]); \
        const uint32_t hs = nlr_dpp_u<SHR + (D)>(head);                \
        const uint32_t ho = r < (D) ? 1u : hs;                         \
        if (!head) {                                                   \
            _Pragma("unroll") for (int c = 0; c < C; ++c) v[c] += vo[c]; \
            head = ho;                                                 \
        }                                                              \
    }
    NLR_MERGE_STEP(1)
    NLR_MERGE_STEP(2)
    NLR_MERGE_STEP(4)
    NLR_MERGE_STEP(8)
#undef NLR_MERGE_STEP
    return valid && tail;
}

// The same scan with the run structure taken from the CELL instead of the address, once for all 8 corners of a point: lanes whose points
// sit in one cell send every corner to one entry, so the head / tail flags and the per-step "add the lane D to the left" masks of
// nlr_run_merge are the same for the 8 corners.  nlr_cell_runs computes them from the integer cell coordinates (exact: equal cells have
// equal corner addresses on every level type; two different cells that collide in a hashed table are simply not merged); nlr_runs_sum
// then costs one DPP add and one select per step and corner.  Same summation tree as nlr_run_merge.
struct NlrRuns {
    bool m1, m2, m4, m8, tail;
};
template <int CTRL>
__device__ __forceinline__ uint32_t nlr_dpp_u0(uint32_t v) {  // out-of-row source: 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float nlr_dpp_f0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ NlrRuns nlr_cell_runs(const uint32_t (&pg)[3], bool valid, int lane) {
    constexpr int SHR = 0x110, SHL = 0x100;
    const int r = lane & 15;
    // invalid lanes never match a neighbour: their first coordinate is replaced by a value no cell has
    const uint32_t k0 = valid ? pg[0] : 0xffffffffu - (uint32_t)lane;
    // (every DPP read is issued unconditionally and lands in a temporary first: inside `&&` / `?:` the compiler may branch around it,
    // and a DPP executed under a partial exec mask reads zeros from the lanes that are switched off)
    const uint32_t l0 = nlr_dpp_u0<SHR + 1>(k0), l1 = nlr_dpp_u0<SHR + 1>(pg[1]), l2 = nlr_dpp_u0<SHR + 1>(pg[2]);
    const bool same = (l0 == k0) & (l1 == pg[1]) & (l2 == pg[2]);
    uint32_t head = ((r < 1) | !same) ? 1u : 0u;
    NlrRuns p;
    const uint32_t right_head = nlr_dpp_u0<SHL + 1>(head);
    p.tail = valid & ((r >= 15) | (right_head != 0u));
#define NLR_PLAN_STEP(D, M)                                             \
    {                                                                   \
        const uint32_t hs = nlr_dpp_u0<SHR + (D)>(head);                \
        const uint32_t ho = r < (D) ? 1u : hs;                          \
        p.M = !head;                                                    \
        head = head ? head : ho;                                        \
    }
    NLR_PLAN_STEP(1, m1)
    NLR_PLAN_STEP(2, m2)
    NLR_PLAN_STEP(4, m4)
    NLR_PLAN_STEP(8, m8)
#undef NLR_PLAN_STEP
    return p;
}
__device__ __forceinline__ float nlr_runs_sum(const NlrRuns &p, float v) {
    constexpr int SHR = 0x110;
    float t;
    t = v + nlr_dpp_f0<SHR + 1>(v); v = p.m1 ? t : v;
    t = v + nlr_dpp_f0<SHR + 2>(v); v = p.m2 ? t : v;
    t = v + nlr_dpp_f0<SHR + 4>(v); v = p.m4 ? t : v;
    t = v + nlr_dpp_f0<SHR + 8>(v); v = p.m8 ? t : v;
    return v;
}
