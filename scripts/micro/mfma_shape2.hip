// Microbenchmark (diagnostic, round 2): why did the 16x16x32 form of the hidden-layer loop take 42.8 cycles per 1 KiB weight
// fragment (floor 32) in round 1's mfma_shape.hip, and what clock does the chip hold on each form?
// Same setting as mfma_shape.hip (one wave per SIMD, 1 KiB A fragment per k-step read from LDS through an 8-deep register ring,
// register-resident bf16 B operands, random data).  Variants:
//   0: 32x32x16, 1 MFMA per fragment                                   (what nlr_mlp_kernel does)
//   1: 16x16x32, 2 MFMAs per fragment, 2 accumulators                  (round 1's variant B)
//   2: 16x16x32, 2 per fragment, 4 accumulators (two fragments interleaved: each accumulator every 4th MFMA)
//   3: 16x16x32, 2 per fragment, fragments stay in registers (no LDS reads)
//   4: variant 0 + a 2-instruction epilogue piece per fragment (v_cvt_pk_bf16_f32 + v_pk_max_i16), as in the real layer
//   5: variant 1 + the same epilogue piece per fragment
//   6: variant 2 + the same epilogue piece per fragment
//   7: 16x16x32, 2 per fragment, 8 accumulators (four fragments interleaved)
// hipcc --offload-arch=gfx950 -O3 -o mfma_shape2 mfma_shape2.hip && ./mfma_shape2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FRAGS 64
template <int V>
__global__ void __launch_bounds__(256, 1) k(const uint4 *__restrict__ w, const uint4 *__restrict__ xin, float *out, int steps,
                                          unsigned long long *clk) {
    __shared__ __align__(16) uint4 lds[FRAGS * 64 + 2560];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < FRAGS * 64; i += 256) lds[i] = w[i];
    __syncthreads();
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i) b[i] = __builtin_bit_cast(bf16x8, xin[(blockIdx.x * 256 + threadIdx.x) * 16 + i]);
    f32x16 a32 = {0};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x16 epsrc;
    for (int i = 0; i < 16; ++i) epsrc[i] = (float)(lane + i) * 0.37f - 9.0f;
    uint32_t epdst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4 ring[8];
    for (int f = 0; f < 8; ++f) ring[f] = lds[f * 64 + lane];
    constexpr bool EPI = (V == 4 || V == 5 || V == 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; s += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint4 fr = ring[u & 7];
            if (V != 3) ring[u & 7] = lds[((s + u + 8) & (FRAGS - 1)) * 64 + lane];
            const bf16x8 av = __builtin_bit_cast(bf16x8, fr);
            if (V == 0 || V == 4) {
                a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[u], a32, 0, 0, 0);
            } else if (V == 1 || V == 3 || V == 5) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
            } else if (V == 2 || V == 6) {
                const int p = (u & 1) * 2;
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[p + 1], 0, 0, 0);
            } else {
                const int p = (u & 3) * 2;
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[p], 0, 0, 0);
                acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[p + 1], 0, 0, 0);
            }
            if (EPI) {
                const f32x2 x = {epsrc[(2 * u) & 15], epsrc[(2 * u + 1) & 15]};
                bf16x2 v = __builtin_convertvector(x, bf16x2);
                const s16x2 z = {0, 0};
                v = __builtin_bit_cast(bf16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
                epdst[u & 7] ^= __builtin_bit_cast(uint32_t, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EPI) epsrc[s & 15] += 1.0f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a32[i];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 4; ++i) r += acc[j][i];
    for (int i = 0; i < 8; ++i) r += (float)epdst[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
#define NV 8
int main() {
    const int blocks = 256, steps = 1 << 19;
    std::vector<uint16_t> h((size_t)FRAGS * 64 * 8), hx((size_t)blocks * 256 * 16 * 8);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    for (auto &v : h) v = rnd();
    for (auto &v : hx) v = rnd();
    uint4 *w, *x; float *out; unsigned long long *clk;
    hipMalloc(&w, h.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    const char *names[NV] = {"0: 32x32x16 1/frag", "1: 16x16x32 2/frag, 2 acc", "2: 16x16x32 2/frag, 4 acc", "3: 16x16x32 2/frag, no LDS",
                             "4: 32x32x16 + epilogue piece", "5: 16x16x32 2 acc + epilogue piece", "6: 16x16x32 4 acc + epilogue piece",
                             "7: 16x16x32 2/frag, 8 acc"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < NV; ++v) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int warm = 0; warm < 6; ++warm) {
                hipEventRecord(a);
                switch (v) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                    default: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps, clk); break;
                }
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long hc[512]; hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            const double flop = (double)blocks * 4 * steps * 32768.0;
            printf("%-40s %.3f ms, %.0f TFLOP/s, in-kernel clock %.0f MHz, %.1f cycles per fragment\n", names[v], ms, flop / ms / 1e9, mhz,
                   (double)hc[0] / steps);
        }
    return 0;
}
