// Round-3 level body of the fused cast + encode kernels (nlr_encode8_kernel / nlr_prop8_kernel): the 8-corner trilinear gather of
// gridencoder.cu:137-197 for ONE multisample point on ONE level, written against the issue cost of the gfx950 vector pipe instead of
// the generic body of nlr_grid_level.h (which stays for the stand-alone operator, the object networks and unusual grids).
//
// What is different from nlr_level_accum (ISA of round 2: ~150 vector instructions per lane and level, of which 42 were the
// 8-lane butterfly, 24 were 64-bit address arithmetic and ~14 a smoothstep that the shipped grids never use):
//   * byte offsets are 32-bit and the level's base is a scalar: every gather is `global_load … v_off, s[base:base+1]`, no 64-bit
//     vector adds, no v_mad_u64_u32;
//   * hashed power-of-two levels: only the low log2(hsize) bits of the products y*P1, z*P2 survive the mask, so the primes are
//     reduced mod hsize on the scalar unit and multiply pre-shifted coordinates (y, z < 2^14, P mod hsize < 2^24: no overflow into
//     the kept bits): offset = ((x<<e) ^ (y<<e)*P1' ^ (z<<e)*P2') & ((hsize-1)<<e), e = log2(bytes per entry); the "+1"
//     corners are adds of (P'<<e); `(a ^ b) & m` is one v_bitop3_b32.  Bit-identical to (x ^ y*P1 ^ z*P2) % hsize;
//   * dense levels: x<<e + y*(step<<e) + z*(step^2<<e), two multiplies (v_mul_lo_u32: full rate on gfx950, the 24-bit multiply is
//     the slower one - profiles/r03_valu_rate_microbench.txt) and adds per corner; entries of <= 8 bytes: the x and x+1 corners are
//     neighbours and come with ONE gather (NlrPair);
//   * waves whose points all sit in one cell fetch its 8 corners through the scalar cache instead (see "wave-uniform cells" below);
//   * linear interpolation and align_corners = False are compile-time (the only values the path uses, grid.py:38-39 defaults);
//   * the sum over the 8 lanes of a multisample group is 3 v_add_f32_dpp per value (inline asm: the compiler pairs the adds into
//     v_pk_add_f32, which cannot carry a DPP operand, and then needs a v_mov_b32_dpp + a zeroing v_mov per operand).
// Arithmetic of the interpolation itself (position, fraction, corner weights (wx*wy)*wz, fmaf accumulation in corner order) is
// unchanged, so features are bit-identical to the generic body's.
#pragma once
#include "nlr_grid_level.h"

typedef float nlr_f2 __attribute__((ext_vector_type(2)));
#ifdef NLR_DBG_ENV
__device__ int nlr_dbg[8] = {0, 99, 0, 0, 0, 0, 0, 0};  // diagnostic builds only, see nlr_encode.hip
#endif

template <typename T, int C>
struct NlrEntry;  // bytes per table entry as a shift, and the gather of one entry from a 32-bit byte offset
template <>
struct NlrEntry<float, 1> {
    static constexpr int E = 2;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[1]) { v[0] = *(const float *)(base + off); }
};
template <>
struct NlrEntry<float, 2> {
    static constexpr int E = 3;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[2]) {
        const float2 t = *(const float2 *)(base + off);
        v[0] = t.x, v[1] = t.y;
    }
};
typedef float nlr_f4 __attribute__((ext_vector_type(4)));
template <>
struct NlrEntry<float, 4> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[4]) {
        const float4 t = *(const float4 *)(base + off);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    }
    // non-temporal form (global_load_dwordx4 ... nt): the line is not kept in L2 at the expense of lines that will be used again
    static __device__ __forceinline__ void ld_nt(const char *base, uint32_t off, float (&v)[4]) {
        const nlr_f4 t = __builtin_nontemporal_load((const nlr_f4 *)(base + off));
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    }
};
template <>
struct NlrEntry<float, 8> {
    static constexpr int E = 5;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[8]) {
        const float4 t = *(const float4 *)(base + off), u = *(const float4 *)(base + off + 16);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w, v[4] = u.x, v[5] = u.y, v[6] = u.z, v[7] = u.w;
    }
};
template <>
struct NlrEntry<__half, 1> {
    static constexpr int E = 1;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[1]) { v[0] = __half2float(*(const __half *)(base + off)); }
};
template <>
struct NlrEntry<__half, 2> {
    static constexpr int E = 2;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[2]) {
        const __half2 t = *(const __half2 *)(base + off);
        v[0] = __low2float(t), v[1] = __high2float(t);
    }
};
template <>
struct NlrEntry<__half, 4> {
    static constexpr int E = 3;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[4]) {
        const uint2 t = *(const uint2 *)(base + off);
        const uint32_t tx = t.x, ty = t.y;  // (named scalars: see NlrPair<__half, 2>)
        const __half2 a = __builtin_bit_cast(__half2, tx), b = __builtin_bit_cast(__half2, ty);
        v[0] = __low2float(a), v[1] = __high2float(a), v[2] = __low2float(b), v[3] = __high2float(b);
    }
};
template <>
struct NlrEntry<__half, 8> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&v)[8]) {
        const uint4 t = *(const uint4 *)(base + off);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 a = __builtin_bit_cast(__half2, w[i]);
            v[2 * i] = __low2float(a), v[2 * i + 1] = __high2float(a);
        }
    }
};

// Dense levels: the x and x + 1 corners are neighbouring entries, so for entries of up to 8 bytes ONE gather instruction fetches both
// (4 gathers per level instead of 8; the pair is only entry-aligned, which global loads allow).
template <typename T, int C>
struct NlrPair {
    static constexpr bool ok = false;
    static __device__ __forceinline__ void ld(const char *, uint32_t, float (&)[C], float (&)[C]) {}
};
typedef float nlr_float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float nlr_float4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t __attribute__((aligned(2))) nlr_u32_a2;
typedef uint32_t nlr_uint2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t nlr_uint4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <>
struct NlrPair<float, 1> {
    static constexpr bool ok = true;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&a)[1], float (&b)[1]) {
        const nlr_float2_a4 t = *(const nlr_float2_a4 *)(base + off);
        a[0] = t.x, b[0] = t.y;
    }
};
template <>
struct NlrPair<float, 2> {
    static constexpr bool ok = true;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&a)[2], float (&b)[2]) {
        const nlr_float4_a8 t = *(const nlr_float4_a8 *)(base + off);
        a[0] = t.x, a[1] = t.y, b[0] = t.z, b[1] = t.w;
    }
};
template <>
struct NlrPair<__half, 1> {
    static constexpr bool ok = true;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&a)[1], float (&b)[1]) {
        const __half2 t = __builtin_bit_cast(__half2, (uint32_t) * (const nlr_u32_a2 *)(base + off));
        a[0] = __low2float(t), b[0] = __high2float(t);
    }
};
template <>
struct NlrPair<__half, 2> {
    static constexpr bool ok = true;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&a)[2], float (&b)[2]) {
        const nlr_uint2_a4 t = *(const nlr_uint2_a4 *)(base + off);
        // (through named scalars: hipcc 7.2 folds `__builtin_bit_cast(__half2, t.y)` on an ext-vector ELEMENT to element 0 - the second
        // entry read as the first; found by the fp16 16 x 2 grid, the first grid with a dense level of 4-byte entry pairs)
        const uint32_t tx = t.x, ty = t.y;
        const __half2 p = __builtin_bit_cast(__half2, tx), q = __builtin_bit_cast(__half2, ty);
        a[0] = __low2float(p), a[1] = __high2float(p), b[0] = __low2float(q), b[1] = __high2float(q);
    }
};
template <>
struct NlrPair<__half, 4> {
    static constexpr bool ok = true;
    static __device__ __forceinline__ void ld(const char *base, uint32_t off, float (&a)[4], float (&b)[4]) {
        const nlr_uint4_a8 t = *(const nlr_uint4_a8 *)(base + off);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __half2 p = __builtin_bit_cast(__half2, w[i]), q = __builtin_bit_cast(__half2, w[2 + i]);
            a[2 * i] = __low2float(p), a[2 * i + 1] = __high2float(p), b[2 * i] = __low2float(q), b[2 * i + 1] = __high2float(q);
        }
    }
};

// Can the fast body run this grid?  (host)  What its 32-bit arithmetic needs, and nothing more (round 3's limits were written for a
// 24-bit multiply the body no longer uses: they sent e.g. every grid finer than 2^14 to the generic kernels):
//   * every level dense (mode 0) or hashed with a power-of-two table (mode 1); linear interpolation, align_corners = False;
//   * byte offsets inside the whole table fit 32 bits (the gathers are `global_load ... v_off, s[base:base+1]` with a 32-bit v_off relative
//     to the LEVEL's base, and the level base itself is added on the scalar unit in 64 bits: the bound is per level);
//   * hashed levels: the kept bits of the wrapped products are the low log2(hsize) + E (E = log2 of the entry's bytes) - multiplication
//     and addition mod 2^32 preserve them exactly when log2(hsize) + E <= 32, which the byte bound above already implies;
//   * dense levels: x + y * step + z * step^2 < hsize (mode 0 by construction), times the entry size below 2^32: the same bound.
static inline bool nlr_level_fast_ok(const GridParams &gp) {
    if (gp.interp != 0 || gp.align_corners != 0) return false;
    const uint64_t entry = (uint64_t)gp.C * (gp.table_dtype == 0 ? 4 : 2);
    for (uint32_t l = 0; l < gp.L; ++l) {
        if (gp.mode[l] > 1) return false;
        if ((uint64_t)gp.hsize[l] * entry > (1ull << 32)) return false;   // byte offsets relative to the level's base
        if (gp.res[l] >= (1u << 30)) return false;                         // coordinate + 1 and its shift by E <= 5 stay exact in 32 bits mod 2^32
    }
    return true;
}

// ---- wave-uniform cells: the 8 corners through the scalar cache ---------------------------------------------------------------------
// The 56 points of a wave are 7 multisamples of 8 CONSECUTIVE samples of one ray; on the coarse levels they mostly sit in one cell
// (bench sweep, NerfMLP grid: 92 / 86 / 78 / 68 / 55 / 40 / 25 / 12 / 3 / 0 % of the waves on levels 0..9).  The vector memory path
// charges one L1 look-up per quad of lanes and instruction whatever the addresses are (TCP_TOTAL_CACHE_ACCESSES = 20 per gather
// instruction = 16 quads x 1.25 lines, and their sum equals the kernel's cycle count: that look-up rate, not arithmetic and not bytes,
// is what bounds the encode kernels), so a wave whose lanes all want the same 8 entries fetches them ONCE with 8 scalar loads
// (scalar cache -> L2: a different path) and feeds them to the FMAs as scalar operands.  Same arithmetic, bit-identical results.
typedef uint32_t nlr_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t nlr_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t nlr_u8 __attribute__((ext_vector_type(8)));
template <int DW>
struct NlrSLoad8;  // 8 entries of DW dwords from base + off[i] (wave-uniform), waited for
#define NLR_SLOAD8(DW, VT, OP)                                                                                                        \
    template <>                                                                                                                       \
    struct NlrSLoad8<DW> {                                                                                                            \
        typedef VT vt;                                                                                                                \
        static __device__ __forceinline__ void ld(const char *base, const uint32_t (&o)[8], VT (&q)[8]) {                             \
            asm volatile(OP " %0, %8, %9\n" OP " %1, %8, %10\n" OP " %2, %8, %11\n" OP " %3, %8, %12\n" OP " %4, %8, %13\n"      \
                         OP " %5, %8, %14\n" OP " %6, %8, %15\n" OP " %7, %8, %16\ns_waitcnt lgkmcnt(0)"                            \
                         : "=&s"(q[0]), "=&s"(q[1]), "=&s"(q[2]), "=&s"(q[3]), "=&s"(q[4]), "=&s"(q[5]), "=&s"(q[6]), "=&s"(q[7])      \
                         : "s"(base), "s"(o[0]), "s"(o[1]), "s"(o[2]), "s"(o[3]), "s"(o[4]), "s"(o[5]), "s"(o[6]), "s"(o[7])           \
                         : "memory");                                                                                                 \
        }                                                                                                                             \
    };
NLR_SLOAD8(1, uint32_t, "s_load_dword")
NLR_SLOAD8(2, nlr_u2, "s_load_dwordx2")
NLR_SLOAD8(4, nlr_u4, "s_load_dwordx4")
NLR_SLOAD8(8, nlr_u8, "s_load_dwordx8")
#undef NLR_SLOAD8
template <int DW>
__device__ __forceinline__ uint32_t nlr_dw(const typename NlrSLoad8<DW>::vt &q, int i) {
    if constexpr (DW == 1) return q;
    else return q[i];
}
// channel c of an entry held as DW scalar dwords
template <typename T, int DW>
__device__ __forceinline__ float nlr_entry_ch(const typename NlrSLoad8<DW>::vt &q, int c) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(float, nlr_dw<DW>(q, c));
    else {
        const __half2 h = __builtin_bit_cast(__half2, nlr_dw<DW>(q, c >> 1));
        return (c & 1) ? __high2float(h) : __low2float(h);
    }
}

// v + s with s wave-uniform, kept an add (the compiler otherwise folds it into the multiply as v_mov + v_mad)
__device__ __forceinline__ uint32_t nlr_add_su(uint32_t v, uint32_t s) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(v));
    return r;
}

// Interpolated value of the level's C channels at (x0, x1, x2) in [0,1]^3, times `werf`.  MODE 0 dense / 1 hashed (wave-uniform).
// Every lane that calls this holds an in-range point (the caller masks the others).
template <typename T, int C, int MODE>
__device__ __forceinline__ void nlr_level_fast(const GridParams &gp, uint32_t level, float x0, float x1, float x2, float werf, float (&a)[C]) {
    constexpr int E = NlrEntry<T, C>::E;
    constexpr int DW = (1 << E) / 4;  // dwords per entry (0: 2-byte entries have no scalar path)
    const float scale = gp.scale[level];
    const char *base = (const char *)gp.table + ((size_t)gp.offset[level] << E);  // scalar
    const float p0 = fmaf(x0, scale, 0.5f), p1 = fmaf(x1, scale, 0.5f), p2 = fmaf(x2, scale, 0.5f);
    // pos >= 0.5: truncation is floor; v_fract_f32 = pos - floor(pos), which is exact in f32 (gridencoder.cu:151-153)
    const uint32_t g0 = (uint32_t)p0, g1 = (uint32_t)p1, g2 = (uint32_t)p2;
    const float f0 = __builtin_amdgcn_fractf(p0), f1 = __builtin_amdgcn_fractf(p1), f2 = __builtin_amdgcn_fractf(p2);
    // corner weights (wx * wy) * wz in the reference's multiplication order, two corners (x = 0 / 1) per packed multiply
    const nlr_f2 wx = {1.0f - f0, f0};
    const float wy0 = 1.0f - f1, wz0 = 1.0f - f2;
    const nlr_f2 wxy0 = wx * wy0, wxy1 = wx * f1;
    const nlr_f2 w2[4] = {wxy0 * wz0, wxy1 * wz0, wxy0 * f2, wxy1 * f2};
    float r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) r[c] = 0.0f;
    // level constants (scalar)
    const uint32_t step = gp.step[level], sy = step << E, sz = (step * step) << E;                 // dense
    const uint32_t hm = gp.hsize[level] - 1u, pa = 2654435761u & hm, pb = 805459861u & hm;          // hashed: the primes mod hsize
    const uint32_t mask = hm << E;
    bool uniform = false;
    uint32_t u0 = 0, u1 = 0, u2 = 0;
    if constexpr (DW >= 1) {
        u0 = __builtin_amdgcn_readfirstlane(g0), u1 = __builtin_amdgcn_readfirstlane(g1), u2 = __builtin_amdgcn_readfirstlane(g2);
        uniform = __builtin_amdgcn_ballot_w64((g0 != u0) | (g1 != u1) | (g2 != u2)) == 0ull;
#ifdef NLR_DBG_ENV
        if (nlr_dbg[3]) uniform = false;
#endif
    }
    if (uniform) {
        if constexpr (DW >= 1) {
            uint32_t so[8];
            if (MODE == 0) {
                const uint32_t ox = u0 << E, oy = u1 * sy, oz = u2 * sz;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) so[c8] = (ox + ((c8 & 1) << E)) + (oy + ((c8 & 2) ? sy : 0u)) + (oz + ((c8 & 4) ? sz : 0u));
            } else {
                const uint32_t hx = u0 << E, hy = (u1 << E) * pa, hz = (u2 << E) * pb;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8)
                    so[c8] = ((hx + ((c8 & 1) << E)) ^ (hy + ((c8 & 2) ? (pa << E) : 0u)) ^ (hz + ((c8 & 4) ? (pb << E) : 0u))) & mask;
            }
            typename NlrSLoad8<DW>::vt q[8];
            NlrSLoad8<DW>::ld(base, so, q);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const float w = (c8 & 1) ? w2[c8 >> 1].y : w2[c8 >> 1].x;
#pragma unroll
                for (int c = 0; c < C; ++c) r[c] = fmaf(w, nlr_entry_ch<T, DW>(q[c8], c), r[c]);
            }
        }
    } else {
        uint32_t off[8];
        if (MODE == 0) {
            const uint32_t ox0 = g0 << E, ox1 = ox0 + (1u << E);
            const uint32_t oy0 = g1 * sy, oy1 = nlr_add_su(oy0, sy);
            const uint32_t oz0 = g2 * sz, oz1 = nlr_add_su(oz0, sz);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) off[c8] = ((c8 & 1) ? ox1 : ox0) + ((c8 & 2) ? oy1 : oy0) + ((c8 & 4) ? oz1 : oz0);
        } else {
            const uint32_t hx0 = g0 << E, hx1 = hx0 + (1u << E);
            const uint32_t hy0 = (g1 << E) * pa, hy1 = nlr_add_su(hy0, pa << E);
            const uint32_t hz0 = (g2 << E) * pb, hz1 = nlr_add_su(hz0, pb << E);
            const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) off[c8] = (((c8 & 1) ? hx1 : hx0) ^ yz[c8 >> 1]) & mask;
        }
        float v[8][C];
        if constexpr (MODE == 0 && NlrPair<T, C>::ok) {
#pragma unroll
            for (int c8 = 0; c8 < 8; c8 += 2) NlrPair<T, C>::ld(base, off[c8], v[c8], v[c8 + 1]);  // off[c8 + 1] = off[c8] + one entry
        } else {
#ifdef NLR_DBG_ENV
            if constexpr (sizeof(T) == 4 && C == 4) {
                if ((nlr_dbg[4] >> level) & 1) {
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) NlrEntry<T, C>::ld_nt(base, off[c8], v[c8]);
                } else {
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) NlrEntry<T, C>::ld(base, off[c8], v[c8]);
                }
            } else
#endif
            {
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) NlrEntry<T, C>::ld(base, off[c8], v[c8]);
            }
        }
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const float w = (c8 & 1) ? w2[c8 >> 1].y : w2[c8 >> 1].x;
#pragma unroll
            for (int c = 0; c < C; ++c) r[c] = fmaf(w, v[c8][c], r[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) a[c] = r[c] * werf;
}

// Sum over the 8 lanes of a multisample group, every lane gets it: xor-1 / xor-2 butterfly steps as quad_perm, the xor-4 step as
// row_half_mirror (after two steps the four lanes of a quad hold the same bits, so lane 7-i delivers what lane i^4 would).  The leading
// s_nop covers the 2 wait states between a VALU write and a DPP read of the same register, which the compiler cannot see through inline
// asm; inside a block the other values' instructions provide them.
#define NLR_DPP_ADD(R, CTRL) "v_add_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf\n"
#define NLR_Q1 "quad_perm:[1,0,3,2]"
#define NLR_Q2 "quad_perm:[2,3,0,1]"
#define NLR_HM "row_half_mirror"
__device__ __forceinline__ void nlr_group8_sum_n(float (&a)[1]) {
    asm volatile("s_nop 1\n" NLR_DPP_ADD("%0", NLR_Q1) "s_nop 1\n" NLR_DPP_ADD("%0", NLR_Q2) "s_nop 1\n" NLR_DPP_ADD("%0", NLR_HM) : "+v"(a[0]));
}
__device__ __forceinline__ void nlr_group8_sum_n(float (&a)[2]) {
    asm volatile("s_nop 1\n" NLR_DPP_ADD("%0", NLR_Q1) NLR_DPP_ADD("%1", NLR_Q1) "s_nop 0\n" NLR_DPP_ADD("%0", NLR_Q2) NLR_DPP_ADD("%1", NLR_Q2)
                 "s_nop 0\n" NLR_DPP_ADD("%0", NLR_HM) NLR_DPP_ADD("%1", NLR_HM)
                 : "+v"(a[0]), "+v"(a[1]));
}
__device__ __forceinline__ void nlr_group8_sum_n(float (&a)[4]) {
    asm volatile("s_nop 1\n" NLR_DPP_ADD("%0", NLR_Q1) NLR_DPP_ADD("%1", NLR_Q1) NLR_DPP_ADD("%2", NLR_Q1) NLR_DPP_ADD("%3", NLR_Q1)
                 NLR_DPP_ADD("%0", NLR_Q2) NLR_DPP_ADD("%1", NLR_Q2) NLR_DPP_ADD("%2", NLR_Q2) NLR_DPP_ADD("%3", NLR_Q2)
                 NLR_DPP_ADD("%0", NLR_HM) NLR_DPP_ADD("%1", NLR_HM) NLR_DPP_ADD("%2", NLR_HM) NLR_DPP_ADD("%3", NLR_HM)
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
}
// The same with the first step out of place: t = sum of r over the group, r is left as it was (the encode kernel keeps r = 0 in lanes
// that hold no point instead of re-zeroing four registers on every level).
#define NLR_DPP_ADD3(D, R, CTRL) "v_add_f32_dpp " D ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf\n"
__device__ __forceinline__ void nlr_group8_sum_to(const float (&r)[4], float (&t)[4]) {
    asm volatile("s_nop 1\n" NLR_DPP_ADD3("%0", "%4", NLR_Q1) NLR_DPP_ADD3("%1", "%5", NLR_Q1) NLR_DPP_ADD3("%2", "%6", NLR_Q1) NLR_DPP_ADD3("%3", "%7", NLR_Q1)
                 NLR_DPP_ADD("%0", NLR_Q2) NLR_DPP_ADD("%1", NLR_Q2) NLR_DPP_ADD("%2", NLR_Q2) NLR_DPP_ADD("%3", NLR_Q2)
                 NLR_DPP_ADD("%0", NLR_HM) NLR_DPP_ADD("%1", NLR_HM) NLR_DPP_ADD("%2", NLR_HM) NLR_DPP_ADD("%3", NLR_HM)
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
}
__device__ __forceinline__ void nlr_group8_sum_n(float (&a)[8]) {
    float lo[4] = {a[0], a[1], a[2], a[3]}, hi[4] = {a[4], a[5], a[6], a[7]};
    nlr_group8_sum_n(lo);
    nlr_group8_sum_n(hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = lo[i], a[4 + i] = hi[i];
}
template <int C>
__device__ __forceinline__ void nlr_group8_sum_to(const float (&r)[C], float (&t)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) t[c] = r[c];
    nlr_group8_sum_n(t);
}
