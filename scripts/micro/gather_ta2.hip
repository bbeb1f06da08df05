// Microbenchmark (diagnostic, round 3): what would sharing a cell's 8 corners inside an 8-lane group buy on the vector L1?
// A wave = 8 groups of 8 lanes (one sample each); every group works on one cell = 8 entries of 16 bytes scattered by a hash.
//   mode 0: today's vector path - every lane loads all 8 entries of its group's cell (8 gather instructions; the 4 lanes of a quad ask
//           for the same entry)
//   mode 1: lane c of a group loads entry c (1 gather instruction; the 4 lanes of a quad ask for 4 different entries)
//   mode 2: as 1, with entries (2k, 2k+1) adjacent (the x / x+1 pair of an even cell or of a dense level)
//   mode 3: all 64 lanes the same 8 entries (wave-uniform cell through the vector path, for reference)
// Prints time per "wave-level" and the implied look-ups if the L1 serves one per clock.
// hipcc --offload-arch=gfx950 -O3 -o gather_ta2 gather_ta2.hip && ./gather_ta2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ t, uint32_t mask, int iters, float *out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t group = MODE == 3 ? (gt >> 6) : (gt >> 3), c = gt & 7;
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const uint32_t cell = h32(group * 977u + it * 0x9e3779b9u);
        if (MODE == 0 || MODE == 3) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = t[h32(cell + u * 0x85ebca6bu) & mask];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        } else {
            uint32_t e = h32(cell + (MODE == 2 ? (c >> 1) : c) * 0x85ebca6bu) & mask;
            if (MODE == 2) e = (e & ~1u) | (c & 1u);
            const float4 v = t[e];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x == 123.456f) out[gt] = acc.y + acc.z + acc.w;
}
int main() {
    for (uint32_t log2 : {16u, 21u}) {  // 1 MiB (L2-resident), 32 MiB (one hashed level of the NerfMLP grid)
        const uint32_t n = 1u << log2;
        float4 *t; float *out;
        (void)hipMalloc(&t, (size_t)n * 16); (void)hipMemset(t, 0, (size_t)n * 16); (void)hipMalloc(&out, 1 << 26);
        const int iters = 64, blocks = 256 * 32;
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double wave_levels = (double)blocks * 4 * iters;             // per launch
            const double cyc = best * 1e-3 * 2.1e9 * 256 / wave_levels;        // CU-cycles per wave-level at 2.1 GHz
            printf("table %3u MiB mode %d: %.3f ms  = %.1f CU-cycles per wave-level\n", (n >> 16), mode, best, cyc);
        }
        (void)hipFree(t); (void)hipFree(out);
    }
    return 0;
}
