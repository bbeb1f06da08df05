// Microbenchmark (diagnostic, informs round 2): the instruction mix of nlr_mlp_kernel's hidden layers - one 1 KiB weight
// fragment per k-step read from LDS by every wave (ds_read_b128, 8-deep read-ahead), bf16 MFMAs on register-resident
// activations, one wave per SIMD, random data - with the two bf16 MFMA shapes:
//   A: v_mfma_f32_32x32x16_bf16, 1 per fragment (32 samples per wave: what the kernel does)
//   C / D: A without LDS reads / with one read per two MFMAs (what a 64-sample-per-wave tiling would do): the LDS share of the power
//   B: v_mfma_f32_16x16x32_bf16, 2 per fragment (two 16-sample column blocks per wave; same FLOPs and LDS bytes per fragment)
// MI355X_MICROARCH.md (DVFS give-back, item 7) says the chip can hold a higher clock on B.  Prints TFLOP/s and the in-kernel clock.
// hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FRAGS 64  // 64 KiB of fragments in LDS, walked round and round
template <int SHAPE>
__global__ void __launch_bounds__(256, 1) k(const uint4 *__restrict__ w, const uint4 *__restrict__ xin, float *out, int steps,
                                          unsigned long long *clk) {
    __shared__ __align__(16) uint4 lds[FRAGS * 64 + 2560];  // + padding so that only one workgroup fits a CU
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < FRAGS * 64; i += 256) lds[i] = w[i];
    __syncthreads();
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i) b[i] = __builtin_bit_cast(bf16x8, xin[(blockIdx.x * 256 + threadIdx.x) * 16 + i]);
    f32x16 a32 = {0};
    f32x4 a16a = {0, 0, 0, 0}, a16b = {0, 0, 0, 0};
    uint4 ring[8];
    for (int f = 0; f < 8; ++f) ring[f] = lds[f * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; s += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint4 fr = ring[u & 7];
            if (SHAPE != 2 && (SHAPE != 3 || (u & 1) == 0)) ring[u & 7] = lds[((s + u + 8) & (FRAGS - 1)) * 64 + lane];
            const bf16x8 av = __builtin_bit_cast(bf16x8, fr);
            if (SHAPE == 0 || SHAPE == 2 || SHAPE == 3) {
                a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[u], a32, 0, 0, 0);
            } else {
                a16a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], a16a, 0, 0, 0);
                a16b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], a16b, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a32[i];
    for (int i = 0; i < 4; ++i) r += a16a[i] + a16b[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256, steps = 1 << 19;
    std::vector<uint16_t> h((size_t)FRAGS * 64 * 8), hx((size_t)blocks * 256 * 16 * 8);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; uint32_t u = __builtin_bit_cast(uint32_t, f); return (uint16_t)(u >> 16); };
    for (auto &v : h) v = rnd();
    for (auto &v : hx) v = rnd();
    uint4 *w, *x; float *out; unsigned long long *clk;
    hipMalloc(&w, h.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 4; ++shape) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int warm = 0; warm < 6; ++warm) {  // a few back-to-back launches so that the clock settles
                hipEventRecord(a);
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk);
                else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk);
                else if (shape == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long hc[512]; hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            const double flop = (double)blocks * 4 * steps * 32768.0;  // per fragment: 32x32x16x2 = 2 x 16x16x32x2
            const char *names[4] = {"32x32x16 (1 per fragment)", "16x16x32 (2 per fragment)", "32x32x16, NO LDS reads (fragments stay in registers)", "32x32x16, one LDS read per TWO MFMAs"};
            printf("%s: %.3f ms, %.0f TFLOP/s, in-kernel clock %.0f MHz, %.1f cycles per fragment\n", names[shape],
                   ms, flop / ms / 1e9, mhz, (double)hc[0] / steps);
        }
    return 0;
}
