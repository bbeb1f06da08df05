// Microbenchmark (diagnostic): cost of 16-byte gathers by address pattern, to learn what the L1/TA path charges for.
//   mode 0: every lane a random 16-B entry                       (1 instruction per entry)
//   mode 1: lane pairs (2i, 2i+1) read ADJACENT entries (e, e^1)  (the x / x+1 corners in one instruction)
//   mode 2: every lane reads e, then e^1 in a second instruction  (what nlr_encode8_kernel does today)
//   mode 3: groups of 8 lanes read the same entry                 (7 multisamples in one cell)
// hipcc --offload-arch=gfx950 -O3 -o gather_ta gather_ta.hip && ./gather_ta
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ t, uint32_t mask, int iters, float *out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t key = MODE == 1 ? (gt >> 1) : MODE == 3 ? (gt >> 3) : gt;
            uint32_t e = h32(key * 977u + (it + u) * 0x9e3779b9u) & mask;
            if (MODE == 1) e = (e & ~1u) | (gt & 1u);
            if (MODE == 2) e = (u & 1) ? ((h32(gt * 977u + (it + u - 1) * 0x9e3779b9u) & mask) ^ 1u) : e;
            v[u] = t[e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 123.456f) out[gt] = acc.y + acc.z + acc.w;
}
int main() {
    for (uint32_t log2 : {16u, 19u, 23u}) {  // 1 MiB (L2), 8 MiB (one hashed level), 128 MiB
        const uint32_t n = 1u << log2;
        float4 *t; float *out;
        hipMalloc(&t, (size_t)n * 16); hipMemset(t, 0, (size_t)n * 16); hipMalloc(&out, 1 << 26);
        const int iters = 256, blocks = 256 * 28;
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(a);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out); break;
                    default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out); break;
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double loads = (double)blocks * 256 * iters;
            printf("table %4u MiB mode %d: %.3f ms  %.1f G lane-loads/s  (%.2f lane-loads/clk/CU at 2.1 GHz)\n", n >> 16, mode, best,
                   loads / best / 1e6, loads / (best * 1e-3) / 256 / 2.1e9);
        }
        hipFree(t); hipFree(out);
    }
    return 0;
}
