// Shared host/device definitions for libnerflidar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nerflidar_hip.h"

#define NLR_EPS 1.1920928955078125e-07f  // torch.finfo(float32).eps
#define NLR_WAVE 64

// ---- error plumbing (thread-local message, int status) ------------------------------------
void nlr_set_error(const char *fmt, ...);
#define NLR_FAIL(code, ...)         \
    do {                            \
        nlr_set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)
#define NLR_CHECK_ARG(cond, ...) \
    do {                         \
        if (!(cond)) NLR_FAIL(NLR_ERR_INVALID, __VA_ARGS__); \
    } while (0)
#define NLR_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            NLR_FAIL(NLR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NLR_LAUNCH_CHECK(name)                                                             \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            NLR_FAIL(NLR_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- per-level grid constants passed by value as kernel arguments --------------------------
struct GridParams {
    const void *table;   // device
    int32_t table_dtype; // 0 f32, 1 f16
    uint32_t L, C;
    uint32_t gridtype, align_corners, interp;
    uint32_t offset[NLR_MAX_GRID_LEVELS];
    uint32_t hsize[NLR_MAX_GRID_LEVELS];
    uint32_t res[NLR_MAX_GRID_LEVELS];    // resolution = ceil(scale)+1 (cu:139)
    float scale[NLR_MAX_GRID_LEVELS];     // exp2f(l*S)*H-1 (cu:138)
    float gsize[NLR_MAX_GRID_LEVELS];     // grid_sizes[l] as float (grid.py:128-129,142)
    float inv_gsize[NLR_MAX_GRID_LEVELS]; // 1.0f / gsize[l] (IEEE division on the host = the device's): a level constant,
                                          // not 12 VALU instructions per lane per level
    uint32_t dense[NLR_MAX_GRID_LEVELS];  // 1 when the dense stride walk never exceeds hsize
    // Index mode per level, decided once on the host (the reference re-derives it per corner, gridencoder.cu:66-84):
    //   0 dense: x + y*s + z*s*s, always < hsize (no modulo);  1 hashed with a power-of-two table: & (hsize-1);
    //   2 generic: the reference's stride walk + modulo, corner by corner
    uint32_t mode[NLR_MAX_GRID_LEVELS];
    uint32_t step[NLR_MAX_GRID_LEVELS];   // s = resolution (+1 unless align_corners)
};

int nlr_fill_grid_params(GridParams *gp, const void *table, int table_dtype, const int32_t *offsets_host,
                         uint32_t L, uint32_t C, float S, uint32_t H, uint32_t gridtype, int align_corners,
                         uint32_t interp);

// ---- device helpers -------------------------------------------------------------------------
#ifdef __HIPCC__
// Index of a corner inside one level (gridencoder.cu:66-84).  D = 3.
__device__ __forceinline__ uint32_t nlr_grid_index(uint32_t gridtype, uint32_t align_corners, uint32_t hsize,
                                                   uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t stride = 1, index = 0;
    const uint32_t step = align_corners ? res : res + 1;
    // unrolled walk over d = 0,1,2 with the `stride <= hashmap_size` guard
    if (stride <= hsize) { index += x * stride; stride *= step; }
    if (stride <= hsize) { index += y * stride; stride *= step; }
    if (stride <= hsize) { index += z * stride; stride *= step; }
    if (gridtype == 0 && stride > hsize) index = x ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % hsize;
}

// Table indices of the 8 cell corners (corner c8: bit d set -> coordinate pg[d]+1), identical to calling
// nlr_grid_index per corner but with the level's mode resolved once and the y/z terms shared between corners.
template <int MODE>
__device__ __forceinline__ void nlr_corner_idx(const GridParams &gp, uint32_t level, const uint32_t (&pg)[3], uint32_t (&idx)[8]) {
    if (MODE == 0) {
        const uint32_t s = gp.step[level];
        const uint32_t y0 = pg[1] * s, z0 = pg[2] * s * s;
        const uint32_t y1 = y0 + s, z1 = z0 + s * s;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) idx[c8] = (pg[0] + (c8 & 1)) + ((c8 & 2) ? y1 : y0) + ((c8 & 4) ? z1 : z0);
    } else if (MODE == 1) {
        const uint32_t mask = gp.hsize[level] - 1u;
        const uint32_t y0 = pg[1] * 2654435761u, z0 = pg[2] * 805459861u;
        const uint32_t y1 = y0 + 2654435761u, z1 = z0 + 805459861u;  // (g+1)*p == g*p + p  (mod 2^32)
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) idx[c8] = ((pg[0] + (c8 & 1)) ^ ((c8 & 2) ? y1 : y0) ^ ((c8 & 4) ? z1 : z0)) & mask;
    } else {
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8)
            idx[c8] = nlr_grid_index(gp.gridtype, gp.align_corners, gp.hsize[level], gp.res[level], pg[0] + (c8 & 1),
                                     pg[1] + ((c8 >> 1) & 1), pg[2] + ((c8 >> 2) & 1));
    }
}

__device__ __forceinline__ float nlr_wave_incl_scan_add(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ float nlr_wave_incl_scan_max(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d, 64);
        if (lane >= d) v = fmaxf(v, o);
    }
    return v;
}
__device__ __forceinline__ float nlr_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float nlr_wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
// torch.nan_to_num(x, nan=0) followed by clip(0,1) (ZI/math.py:106)
__device__ __forceinline__ float nlr_nan0_clip01(float x) {
    if (x != x) x = 0.0f;
    return fminf(fmaxf(x, 0.0f), 1.0f);
}
#endif
