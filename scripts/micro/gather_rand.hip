// Microbenchmark (diagnostic, round 4): the chip's rate for RANDOM 16-byte gathers - the access pattern of the hashed fine levels of
// the NerfMLP grid on a trained scene, where every multisample point of a wave sits in its own cell (tests: profiles/r04_*): each
// lane reads the 8 corners of its own random cell = 4 random x-pairs (two adjacent 16-byte entries in one 64-byte line).
//   mode 0: 8 gathers per lane and "level" (both entries of each pair: what nlr_level_fast issues), consumed level by level
//   mode 1: 4 gathers per lane and level (one entry per pair): what the second access to a line costs
//   mode 2: as 0 with TWO levels in flight (16 gathers before the first use)
//   mode 3: as 0 with FOUR levels in flight (32 gathers)
// over tables of 32 MiB (one hashed level), 256 MiB (the NerfMLP table + the proposal tables) and 1 GiB (beyond the Infinity Cache), all CUs busy.
// Prints time, distinct 64-byte lines per second chip-wide and CU-cycles per line at 2.1 GHz.
// hipcc --offload-arch=gfx950 -O3 -o gather_rand gather_rand.hip && ./gather_rand
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int PER, int DEPTH>  // PER entries per pair read (1 or 2), DEPTH levels in flight
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ t, uint32_t mask, int iters, float *out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += DEPTH) {
        float4 v[DEPTH][4][PER];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint32_t cell = h32(gt * 977u + (it + d) * 0x9e3779b9u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = h32(cell + u * 0x85ebca6bu) & mask & ~1u;
#pragma unroll
                for (int p = 0; p < PER; ++p) v[d][u][p] = t[e | p];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < PER; ++p) { acc.x += v[d][u][p].x; acc.y += v[d][u][p].y; acc.z += v[d][u][p].z; acc.w += v[d][u][p].w; }
    }
    if (acc.x == 123.456f) out[gt] = acc.y + acc.z + acc.w;
}
int main() {
    const size_t mibs[3] = {32, 256, 1024};
    for (size_t mib : mibs) {
        size_t n = mib << 16;  // 16-byte entries
        uint32_t pow2 = 1;
        while ((size_t)pow2 * 2 <= n) pow2 *= 2;
        float4 *t; float *out;
        (void)hipMalloc(&t, n * 16); (void)hipMemset(t, 0, n * 16); (void)hipMalloc(&out, 1 << 26);
        const int iters = 32, blocks = 256 * 64;
        for (int mode = 0; mode < 4; ++mode) {
            // the mask covers the largest power of two <= the table; for 229 MiB a second pass folds the index so that the whole table is hit
            const uint32_t mask = pow2 - 1;
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL((k<2, 1>), dim3(blocks), dim3(256), 0, 0, t, mask, iters, out);
                if (mode == 1) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), 0, 0, t, mask, iters, out);
                if (mode == 2) hipLaunchKernelGGL((k<2, 2>), dim3(blocks), dim3(256), 0, 0, t, mask, iters, out);
                if (mode == 3) hipLaunchKernelGGL((k<2, 4>), dim3(blocks), dim3(256), 0, 0, t, mask, iters, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double lines = (double)blocks * 256 * iters * 4;  // distinct 64-byte lines per launch (pairs)
            printf("table %4zu MiB (indexed %4u MiB) mode %d: %.3f ms  %.1f G lines/s = %.2f TB/s of 64-byte lines, %.2f CU-cycles per line\n", mib,
                   (unsigned)(pow2 >> 16), mode, best, lines / best / 1e6, lines * 64 / best / 1e9, best * 1e-3 * 2.1e9 * 256 / lines);
        }
        (void)hipFree(t); (void)hipFree(out);
    }
    return 0;
}
