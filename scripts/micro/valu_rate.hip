// Microbenchmark (diagnostic, round 3): issue cost of the vector instructions the hash-grid level body is made of, per SIMD, at 1 / 2 / 4 / 8
// waves per SIMD.  The round-2 verdict asked for an issue-CYCLE account of nlr_encode8_kernel instead of an instruction count: which ops are
// full rate (4 cycles per wave64 instruction), which are not (v_mul_lo_u32, transcendentals, DPP), and whether packed f32 (v_pk_fma_f32 /
// v_pk_mul_f32) really does two lanes' worth of work per issue slot on gfx950.
// Every op runs as 8 independent dependency chains x 8 instructions per loop body (64 instructions), `iters` bodies per wave.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one kernel per op: OPSTR uses %0 as the chain register (read-modify-write), %1 / %2 as loop-invariant operands
#define DEF_KERNEL_1(NAME, OPSTR)                                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, unsigned long long *clk) {                                  \
        float r[8];                                                                                                                \
        for (int i = 0; i < 8; ++i) r[i] = (float)(threadIdx.x + i) * 0.001f + 1.0f;                                               \
        float a = 1.0000001f, b = 1e-9f;                                                                                           \
        asm volatile("" : "+v"(a), "+v"(b));                                                                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                        \
                asm volatile(OPSTR "\n" : "+v"(r[0]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[1]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[2]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[3]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[4]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[5]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[6]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[7]) : "v"(a), "v"(b));                                                            \
            }                                                                                                                      \
        }                                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s = 0;                                                                                                               \
        for (int i = 0; i < 8; ++i) s += r[i];                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                            \
        if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                  \
    }

// packed variants: chain registers are 64-bit pairs
#define DEF_KERNEL_2(NAME, OPSTR)                                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, unsigned long long *clk) {                                  \
        f32x2 r[8];                                                                                                                \
        for (int i = 0; i < 8; ++i) r[i] = f32x2{(float)(threadIdx.x + i) * 0.001f + 1.0f, 0.5f};                                  \
        f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-9f, 2e-9f};                                                                    \
        asm volatile("" : "+v"(a), "+v"(b));                                                                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                        \
                asm volatile(OPSTR "\n" : "+v"(r[0]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[1]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[2]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[3]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[4]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[5]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[6]) : "v"(a), "v"(b));                                                            \
                asm volatile(OPSTR "\n" : "+v"(r[7]) : "v"(a), "v"(b));                                                            \
            }                                                                                                                      \
        }                                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s = 0;                                                                                                               \
        for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                            \
        if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                  \
    }

DEF_KERNEL_1(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_1(k_mul, "v_mul_f32 %0, %0, %1")
DEF_KERNEL_1(k_add, "v_add_f32 %0, %0, %2")
DEF_KERNEL_2(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_2(k_pk_fma_sel, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]")
DEF_KERNEL_2(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL_2(k_pk_add, "v_pk_add_f32 %0, %0, %2")
DEF_KERNEL_1(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_1(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL_1(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL_1(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL_1(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL_1(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEF_KERNEL_1(k_lshl_add, "v_lshl_add_u32 %0, %0, 4, %2")
DEF_KERNEL_1(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL_1(k_bfe, "v_bfe_u32 %0, %0, 4, 21")
DEF_KERNEL_1(k_exp, "v_exp_f32 %0, %0")
DEF_KERNEL_1(k_rcp, "v_rcp_f32 %0, %0")
DEF_KERNEL_1(k_rsq, "v_rsq_f32 %0, %0")
DEF_KERNEL_1(k_floor, "v_floor_f32 %0, %0")
DEF_KERNEL_1(k_fract, "v_fract_f32 %0, %0")
DEF_KERNEL_1(k_cvt_u32, "v_cvt_u32_f32 %0, %0")
DEF_KERNEL_1(k_cvt_f32, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL_1(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL_1(k_add_dpp_quad, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL_1(k_add_dpp_hmirror, "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf")
DEF_KERNEL_1(k_add_dpp_shr, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL_1(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL_1(k_max3, "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL_1(k_cvt_f16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL_1(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL_1(k_mov, "v_mov_b32 %0, %1")

typedef void (*kfn)(float *, int, unsigned long long *);
struct Op {
    const char *name;
    kfn fn;
    int lanes_work;  // f32 results per lane per instruction (2 for packed)
};

int main() {
    Op ops[] = {{"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_pk_fma_f32", k_pk_fma, 2},
                {"v_pk_fma_f32 op_sel bcast", k_pk_fma_sel, 2}, {"v_pk_mul_f32", k_pk_mul, 2}, {"v_pk_add_f32", k_pk_add, 2},
                {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_xor_b32", k_xor, 1},
                {"v_and_or_b32", k_and_or, 1}, {"v_xad_u32", k_xad, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_add3_u32", k_add3, 1},
                {"v_bfe_u32", k_bfe, 1}, {"v_exp_f32", k_exp, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_floor_f32", k_floor, 1},
                {"v_fract_f32", k_fract, 1}, {"v_cvt_u32_f32", k_cvt_u32, 1}, {"v_cvt_f32_u32", k_cvt_f32, 1}, {"v_cndmask_b32", k_cndmask, 1},
                {"v_add_f32_dpp quad_perm", k_add_dpp_quad, 1}, {"v_add_f32_dpp row_half_mirror", k_add_dpp_hmirror, 1},
                {"v_add_f32_dpp row_shr:1", k_add_dpp_shr, 1}, {"v_mov_b32_dpp quad_perm", k_mov_dpp, 1}, {"v_max3_f32", k_max3, 1},
                {"v_cvt_f32_f16", k_cvt_f16, 1}, {"v_perm_b32", k_perm, 1}, {"v_mov_b32", k_mov, 1}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 2000;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipMalloc(&clk, (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(cus * 8 * 4);
    printf("# %s, %d CUs; cycles per wave64 instruction per SIMD = median wave cycles / (instructions per wave x waves per SIMD)\n", prop.gcnArchName, cus);
    printf("%-34s %10s %10s %10s %10s   (wave-alone cycles/instr)\n", "op", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (const Op &op : ops) {
        double res[4], alone = 0;
        int wi = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;  // 256-thread blocks = 4 waves = one per SIMD
            hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, out, 10, clk);
            hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), clk, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<unsigned long long> v(h.begin(), h.begin() + blocks * 4);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            const double per = med / ((double)iters * 64.0);
            if (wps == 1) alone = per;
            res[wi++] = per / wps;
        }
        printf("%-34s %10.2f %10.2f %10.2f %10.2f   (%.2f)\n", op.name, res[0], res[1], res[2], res[3], alone);
    }
    return 0;
}
