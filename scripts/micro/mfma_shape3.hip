// Microbenchmark (diagnostic, round 2): placement of the 1 KiB LDS fragment read relative to the two 16x16x32 MFMAs that consume a
// fragment.  mfma_shape2 showed 42.8 cycles per fragment with the read between them and 33.5 without any read.
// hipcc --offload-arch=gfx950 -O3 -o mfma_shape3 mfma_shape3.hip && ./mfma_shape3
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
    constexpr bool EPI = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; s += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // V: 0 = [mfma0, ds_read, mfma1] (round-2a order)   1 = [ds_read, mfma0, mfma1]   2 = [mfma0, mfma1, ds_read]
            //    3 = ds_read_b64 x2 instead of b128, order 1       4 = 4 MFMAs per fragment (64 samples per wave), order 1
            //    5 = two fragments per step: [ds_read, ds_read, 4 mfma]    6 = order 1 + s_nop 1 after the read    7 = 32x32x16 reference
            const uint4 fr = ring[u & 7];
            const bf16x8 av = __builtin_bit_cast(bf16x8, fr);
            const int nx = ((s + u + 8) & (FRAGS - 1)) * 64 + lane;
            if (V == 0) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                ring[u & 7] = lds[nx];
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
            } else if (V == 1 || V == 6) {
                ring[u & 7] = lds[nx];
                if (V == 6) asm volatile("s_nop 1");
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
            } else if (V == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
                ring[u & 7] = lds[nx];
            } else if (V == 3) {
                const uint2 *l2 = reinterpret_cast<const uint2 *>(lds);
                const uint2 p0 = l2[nx * 2], p1 = l2[nx * 2 + 1];
                ring[u & 7] = uint4{p0.x, p0.y, p1.x, p1.y};
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
            } else if (V == 4) {
                ring[u & 7] = lds[nx];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 3) & 15], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 9) & 15], acc[3], 0, 0, 0);
            } else if (V == 5) {
                if ((u & 1) == 0) {
                    const uint4 fr2 = ring[(u + 1) & 7];
                    const bf16x8 av2 = __builtin_bit_cast(bf16x8, fr2);
                    ring[u & 7] = lds[nx];
                    ring[(u + 1) & 7] = lds[((s + u + 9) & (FRAGS - 1)) * 64 + lane];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(u + 5) & 15], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av2, b[(u + 1) & 15], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av2, b[(u + 6) & 15], acc[3], 0, 0, 0);
                }
            } else {
                ring[u & 7] = lds[nx];
                a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[u], a32, 0, 0, 0);
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
    const char *names[NV] = {"0: 16x16x32 [mfma, read, mfma]", "1: 16x16x32 [read, mfma, mfma]", "2: 16x16x32 [mfma, mfma, read]", "3: 16x16x32 read as 2 x b64",
                             "4: 16x16x32 4 MFMA/frag (64 samples)", "5: 16x16x32 [read, read, 4 mfma]", "6: 16x16x32 [read, nop, mfma, mfma]", "7: 32x32x16 [read, mfma]"};
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
            const double flop = (double)blocks * 4 * steps * 32768.0 * (v == 4 ? 2.0 : 1.0);
            printf("%-40s %.3f ms, %.0f TFLOP/s, in-kernel clock %.0f MHz, %.1f cycles per fragment\n", names[v], ms, flop / ms / 1e9, mhz,
                   (double)hc[0] / steps);
        }
    return 0;
}
