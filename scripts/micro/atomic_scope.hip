// Microbenchmark (diagnostic, round 4): where does a float atomic add execute, and what does it cost?  The hash-grid backward
// (nlr_grid_bwd_kernel<4>) sends 16-byte entries' worth of `global_atomic_add_f32` (4 lanes on 4 adjacent floats) to random rows of
// 32 MiB tables.  An MI355X has 8 XCDs with one L2 each; an agent-scope atomic must be visible to the other seven.
//   mode 0: agent scope (plain atomicAdd), ONE table                        = what the kernel does today
//   mode 1: agent scope, one table copy per XCD (copy chosen by HW_REG_XCC_ID): the effect of the copies alone
//   mode 2: workgroup scope, one table copy per XCD: the L2 of the XCD may keep the line and own the sum
//   mode 3: wavefront scope, one copy per XCD
//   mode 4: workgroup scope, ONE table (expected to lose updates across XCDs: printed as a check of the premise, not a candidate)
// `window` = number of entries the random rows are drawn from (32 MiB = one hashed level; 1 MiB ~ what an L2 holds with room to spare).
// Every add is 1.0f, so sum(tables) must equal the number of adds: printed as `sum ok`.
// hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip && ./atomic_scope
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
#define MY_XCC() (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7)  // HW_REG_XCC_ID, bits 3:0

template <int MODE>
__global__ void __launch_bounds__(256) k(float *__restrict__ t, uint32_t mask, size_t copy_floats, int iters, uint32_t *xcc_seen) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, c = lane >> 4;             // channel-major rows: 16 points x 4 channels per wave
    const uint32_t pt = (gt >> 6) * 16 + (lane & 15);
    const uint32_t xcc = MY_XCC();
    if (threadIdx.x == 0) atomicOr(xcc_seen + xcc, 1u);
    float *base = t + ((MODE == 0 || MODE == 4) ? 0 : (size_t)xcc * copy_floats);
    for (int it = 0; it < iters; ++it) {
        const uint32_t e = h32(pt * 977u + it * 0x9e3779b9u) & mask;
        float *p = base + (size_t)e * 4 + c;
        if (MODE == 0 || MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2 || MODE == 4) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
// Second table: what is the unit of cost - the lane, the 64-byte line, the instruction, the CU?  `span` lanes share one 64-byte line
// (span 1: every lane its own random line, one float; 4: the kernel's layout, 4 lanes on one 16-byte entry; 16: four adjacent entries;
// 64 with same = true: all lanes one address), `active` lanes of 64 issue the add.
__global__ void __launch_bounds__(256) k2(float *__restrict__ t, uint32_t mask, int span, int active, int same, int iters) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t grp = (gt >> 6) * 64 + lane / span * span;          // lanes of a group draw the same line
    if ((int)lane >= active) return;
    for (int it = 0; it < iters; ++it) {
        const uint32_t line = h32(grp * 977u + it * 0x9e3779b9u) & mask;        // mask counts 64-byte lines
        float *p = t + (size_t)line * 16 + (same ? 0 : lane % span);
        __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void total(const float *t, size_t n, double *out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += t[i];
    for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
int main() {
    const size_t entries = 2u << 20;                 // 2^21 rows x 16 bytes = 32 MiB: one hashed level of the NerfMLP grid
    const size_t copy_floats = entries * 4;
    float *t; double *sum; uint32_t *seen;
    (void)hipMalloc(&t, 8 * copy_floats * 4); (void)hipMalloc(&sum, 8); (void)hipMalloc(&seen, 64);
    const int iters = 64, blocks = 256 * 32;        // 2.1 M threads x 64 adds = 134 M float adds = 33.5 M entry updates per launch
    const double adds = (double)blocks * 256 * iters;
    const uint32_t windows[3] = {(uint32_t)entries, 1u << 16, 1u << 12};
    for (uint32_t w : windows) {
        printf("rows drawn from %u entries (%.2f MiB per table copy)\n", w, w * 16.0 / 1048576);
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            float best = 1e9f; double got = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(t, 0, 8 * copy_floats * 4); (void)hipMemset(sum, 0, 8); (void)hipMemset(seen, 0, 64);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, t, w - 1, copy_floats, iters, seen);
                if (mode == 1) hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(256), 0, 0, t, w - 1, copy_floats, iters, seen);
                if (mode == 2) hipLaunchKernelGGL((k<2>), dim3(blocks), dim3(256), 0, 0, t, w - 1, copy_floats, iters, seen);
                if (mode == 3) hipLaunchKernelGGL((k<3>), dim3(blocks), dim3(256), 0, 0, t, w - 1, copy_floats, iters, seen);
                if (mode == 4) hipLaunchKernelGGL((k<4>), dim3(blocks), dim3(256), 0, 0, t, w - 1, copy_floats, iters, seen);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                hipLaunchKernelGGL(total, dim3(1024), dim3(256), 0, 0, t, 8 * copy_floats, sum);
                (void)hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost);
            }
            uint32_t hs[16]; (void)hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost);
            int nx = 0; for (int i = 0; i < 8; ++i) nx += hs[i] != 0;
            const char *names[5] = {"agent scope, one table      ", "agent scope, copy per XCD   ", "workgroup scope, copy per XCD", "wavefront scope, copy per XCD", "workgroup scope, ONE table   "};
            printf("  mode %d %s: %7.3f ms, %6.1f G float adds/s = %5.1f G entries/s, sum %s (%.0f of %.0f), XCC ids seen %d\n", mode, names[mode], best,
                   adds / best * 1e-6, adds / 4 / best * 1e-6, got == adds ? "ok" : "WRONG", got, adds, nx);
        }
    }
    printf("cost model (32 MiB table; float adds by `active` lanes of a wave, `span` lanes per 64-byte line)\n");
    struct V { int span, active, same, blocks; const char *what; } vs[] = {
        {1, 64, 0, 8192, "64 lanes, 64 lines"}, {4, 64, 0, 8192, "64 lanes, 16 lines (one entry per 4 lanes)"}, {16, 64, 0, 8192, "64 lanes, 4 lines"},
        {64, 64, 0, 8192, "64 lanes, 1 line, 64... 16 distinct floats"}, {64, 64, 1, 8192, "64 lanes, one address"}, {1, 16, 0, 8192, "16 lanes, 16 lines"},
        {4, 16, 0, 8192, "16 lanes, 4 lines"}, {1, 64, 0, 256, "64 lanes, 64 lines, 256 workgroups (one per CU)"}, {1, 64, 0, 64, "64 lanes, 64 lines, 64 workgroups"},
        {4, 64, 0, 64, "64 lanes, 16 lines, 64 workgroups"}};
    for (V v : vs) {
        const int it2 = v.blocks >= 8192 ? 64 : 64 * 16;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k2, dim3(v.blocks), dim3(256), 0, 0, t, (uint32_t)(entries / 4 - 1), v.span, v.active, v.same, it2);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double insts = (double)v.blocks * 4 * it2, lanes = insts * v.active;
        printf("  %-50s: %7.3f ms, %6.1f G lane adds/s, %6.2f G wave instructions/s\n", v.what, best, lanes / best * 1e-6, insts / best * 1e-6);
    }
    return 0;
}
