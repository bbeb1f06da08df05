// Point cloud -> spherical range image (the glue between the renderer's points/labels/rgb and the ray-drop UNet).
//
// Replaces NeRF_Lidar_code/src/lidar_utils.py:215-282 (LaserScan.do_range_projection, numpy on the host: argsort by
// depth, far -> near painter's scatter so that the nearest point of a pixel is written last).  Here the "nearest
// wins" rule is a 64-bit atomicMin on the IEEE bits of the (positive, double) depth per pixel, then an atomicMin on
// the point index among the points that hit that minimum (ties: smallest index, what a stable sort would give), then
// one gather pass per pixel.  All angle arithmetic is double, as in the reference (its inputs are float64 after
// nerf2world.py:22-38), so pixel assignment is bit-identical to the numpy result.
// Quirk kept: proj_mask = (proj_idx > 0), i.e. the point with index 0 never counts as a hit (lidar_utils.py:281).
#include "nlr_common.h"

struct RangeParams {
    const double *points;  // [N,3] in the LiDAR frame
    const float *semantic; // [N] or null
    const float *rgb;      // [N,3] or null
    uint32_t N, H, W;
    double fov_up, fov_down;  // radians
    unsigned long long *best_depth;  // [H*W] workspace
    unsigned int *best_idx;          // [H*W] workspace
    float *proj_range, *proj_xyz, *proj_semantic, *proj_rgb, *proj_mask;
    int32_t *proj_idx;
};

__device__ __forceinline__ bool nlr_range_cell(const RangeParams &P, uint32_t i, uint32_t &cell, double &depth) {
    const double x = P.points[(size_t)i * 3], y = P.points[(size_t)i * 3 + 1], z = P.points[(size_t)i * 3 + 2];
    depth = sqrt(x * x + y * y + z * z);
    const double fov = fabs(P.fov_down) + fabs(P.fov_up);
    const double yaw = -atan2(y, x);
    const double pitch = asin(z / depth);
    double px = 0.5 * (yaw / M_PI + 1.0);
    double py = 1.0 - (pitch + fabs(P.fov_down)) / fov;
    px *= (double)P.W;
    py *= (double)P.H;
    px = fmax(0.0, fmin((double)P.W - 1.0, floor(px)));
    py = fmax(0.0, fmin((double)P.H - 1.0, floor(py)));
    // NaN (zero-length point: asin(0/0)) compares false everywhere; numpy's astype(int32) of NaN is undefined -> skip
    if (!(px == px) || !(py == py) || !(depth == depth)) return false;
    cell = (uint32_t)py * P.W + (uint32_t)px;
    return true;
}

__global__ void __launch_bounds__(256) nlr_range_init_kernel(RangeParams P) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.H * P.W) return;
    P.best_depth[c] = ~0ull;
    P.best_idx[c] = ~0u;
}
__global__ void __launch_bounds__(256) nlr_range_depth_kernel(RangeParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    uint32_t cell;
    double depth;
    if (!nlr_range_cell(P, i, cell, depth)) return;
    atomicMin(&P.best_depth[cell], (unsigned long long)__double_as_longlong(depth));
}
__global__ void __launch_bounds__(256) nlr_range_index_kernel(RangeParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    uint32_t cell;
    double depth;
    if (!nlr_range_cell(P, i, cell, depth)) return;
    if ((unsigned long long)__double_as_longlong(depth) == P.best_depth[cell]) atomicMin(&P.best_idx[cell], i);
}
__global__ void __launch_bounds__(256) nlr_range_gather_kernel(RangeParams P) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.H * P.W) return;
    const unsigned int i = P.best_idx[c];
    const bool hit = i != ~0u;
    if (P.proj_range) P.proj_range[c] = hit ? (float)__longlong_as_double((long long)P.best_depth[c]) : -1.0f;
    if (P.proj_xyz)
        for (int k = 0; k < 3; ++k) P.proj_xyz[(size_t)c * 3 + k] = hit ? (float)P.points[(size_t)i * 3 + k] : -1.0f;
    if (P.proj_semantic) P.proj_semantic[c] = hit ? (P.semantic ? P.semantic[i] : 0.0f) : -1.0f;
    if (P.proj_rgb)
        for (int k = 0; k < 3; ++k) P.proj_rgb[(size_t)c * 3 + k] = (hit && P.rgb) ? P.rgb[(size_t)i * 3 + k] : 0.0f;
    if (P.proj_idx) P.proj_idx[c] = hit ? (int32_t)i : -1;
    if (P.proj_mask) P.proj_mask[c] = (hit && i > 0) ? 1.0f : 0.0f;
}

extern "C" size_t nlr_range_workspace_bytes(uint32_t H, uint32_t W) { return (size_t)H * W * (sizeof(unsigned long long) + sizeof(unsigned int)); }

extern "C" int nlr_range_project(const double *points, const float *semantic, const float *rgb, uint32_t N, uint32_t H, uint32_t W,
                                 float fov_up_deg, float fov_down_deg, void *workspace, size_t workspace_bytes, float *proj_range,
                                 float *proj_xyz, float *proj_semantic, float *proj_rgb, int32_t *proj_idx, float *proj_mask,
                                 void *stream) {
    NLR_CHECK_ARG((points || N == 0) && H > 0 && W > 0, "range_project: NULL points or empty image");
    if (!workspace || workspace_bytes < nlr_range_workspace_bytes(H, W))
        NLR_FAIL(NLR_ERR_WORKSPACE, "range_project: workspace %zu B < %zu B", workspace_bytes, nlr_range_workspace_bytes(H, W));
    RangeParams P;
    memset(&P, 0, sizeof(P));
    P.points = points;
    P.semantic = semantic;
    P.rgb = rgb;
    P.N = N;
    P.H = H;
    P.W = W;
    P.fov_up = (double)fov_up_deg / 180.0 * M_PI;
    P.fov_down = (double)fov_down_deg / 180.0 * M_PI;
    P.best_depth = (unsigned long long *)workspace;
    P.best_idx = (unsigned int *)((char *)workspace + (size_t)H * W * sizeof(unsigned long long));
    P.proj_range = proj_range;
    P.proj_xyz = proj_xyz;
    P.proj_semantic = proj_semantic;
    P.proj_rgb = proj_rgb;
    P.proj_idx = proj_idx;
    P.proj_mask = proj_mask;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gc((H * W + 255) / 256), gp((N + 255) / 256), b(256);
    hipLaunchKernelGGL(nlr_range_init_kernel, gc, b, 0, st, P);
    if (N) {
        hipLaunchKernelGGL(nlr_range_depth_kernel, gp, b, 0, st, P);
        hipLaunchKernelGGL(nlr_range_index_kernel, gp, b, 0, st, P);
    }
    hipLaunchKernelGGL(nlr_range_gather_kernel, gc, b, 0, st, P);
    NLR_LAUNCH_CHECK("nlr_range_*_kernel");
    return NLR_OK;
}
