// NerfMLP evaluation on the matrix cores: density trunk -> semantic/intensity heads -> view MLP -> rgb.
//
// Replaces (rows a-9..a-12 of the scope table):
//   ZI/models.py:887-889, 996-997, 1116     density_layer (F->64->256), softplus(raw - 1)
//   ZI/models.py:954-961, 1124-1143         sem_layer (256->64->19, softmax), intensity_layer (256->64->1)
//   ZI/coord.py:199-210, models.py:1190-1196  pos_enc(viewdirs) broadcast over samples
//   ZI/models.py:939-951, 1223-1234, 1251   lin_second_stage_i (+skip concat after layer 0), rgb_layer, sigmoid, padding
//
// Design (CDNA4, not a translation of the nn.Linear chain):
//   * the whole chain runs TRANSPOSED, activations^T = W . x^T, so that an MFMA result tile (32 output
//     features x 32 samples: sample on the lane, features in the 16 accumulator registers) is already
//     the B operand of the next layer's MFMA (cdna_hip_programming.md section 3, "An accumulator tile as
//     the next MFMA's operand").  Activations never leave the register file: no LDS round trip, no
//     barrier between the 8+ layers.  One wavefront owns 32 samples end to end.
//   * weights are the A operand, pre-packed at model-create time into exactly the per-lane fragment
//     order (including the permuted k order the accumulator layout implies), so every fragment fetch is
//     one fully coalesced 16-byte-per-lane load of 1 KiB per wavefront.
//   * the 27 direction-encoding features are computed once per ray by a small pre-kernel and ride through view
//     layers 0 and 1 as one extra zero-padded 32-feature input tile, so every GEMM has K % 32 == 0;
//   * nothing but the weight tape is read from global memory after the prologue: all biases sit in LDS.
//   * precision: layers whose error reaches depth / semantic argmax / intensity (density trunk, heads)
//     use the exact-f32 MFMA (v_mfma_f32_32x32x2_f32); the view MLP (rgb only, 92 % of the MACs) uses
//     bf16 MFMA (v_mfma_f32_32x32x16_bf16) with f32 accumulation.  NLR_PREC_F32 runs everything in f32.
#pragma once
#include "nlr_kernels.h"

#include <type_traits>


// row of accumulator register r for lane half h inside a 32-row tile
__device__ __forceinline__ int nlr_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- weight tape: global -> LDS by LDS-DMA (triple buffered), shared by the 4 waves of a workgroup ----------------
// A chunk is 32 KiB = 32 fragments of 1 KiB (one fragment = the A operand of one MFMA for all 64 lanes).  Every
// wave needs every fragment (each wave owns 32 samples and all output features), so staging through LDS cuts the
// L2 -> CU weight traffic 4x against per-wave global loads and puts the fragment reads on ds_read_b128.
// The refill is `global_load_lds_dwordx4` (one fragment = one wave-instruction, lane-linear in LDS, which is the
// fragment layout): no VGPR staging and no ds_write pass.  Measured with in-kernel stamps, the
// register-staged refill (8 global_load + 8 ds_write_b128 per wave per chunk) cost 38 % of the kernel: the
// VGPR -> LDS store path moves ~79 B/clk/CU and does not overlap the fragment reads.
// Schedule inside chunk c (f = fragment position, all positions are compile-time after unrolling):
//   f = 8   s_waitcnt vmcnt(0) (this wave's quarter of chunk c+1, requested at (16, c-1), has landed); signal
//   f = 16  once all four waves have signalled, each wave requests its quarter (8 fragments) of chunk c+2 into LDS
//           buffer (c+2)%3 = the buffer of chunk c-1 (hand-shake: see Tape::signal)
//   every f: the fragment f+8 is requested into an 8-deep register ring right after fragment f is consumed; from
//            f = 24 on these requests run into chunk c+1, so no LDS latency is exposed at a chunk boundary.
// The last ds_read of chunk c-1 (issued at (23, c-1)) was waited for by the MFMA that consumed it at (7, c) and LDS
// reads return in order, so at the signal every read of chunk c-1 by this wave is complete; the DMA is tracked by vmcnt.
#define NLR_CHUNK_FRAGS 32                       // fragments (1 KiB each) per chunk
#define NLR_CHUNK_SLOTS (NLR_CHUNK_FRAGS * 64)   // uint4 slots per chunk
#define NLR_NBUF 3
#ifndef NLR_SIG_F
#define NLR_SIG_F 8
#endif
#ifndef NLR_POLL_F
#define NLR_POLL_F 22
#endif
#ifndef NLR_PF
#define NLR_PF 8                                 // fragment read-ahead (register ring)
#endif
typedef const __attribute__((address_space(1))) void *nlr_gptr;
typedef __attribute__((address_space(3))) void *nlr_lptr;
struct Tape {
    const uint4 *__restrict__ base;
    uint4 *lds;  // [NLR_NBUF][NLR_CHUNK_SLOTS]
    uint4 ring[NLR_PF];
    // cur: chunks consumed so far by this workgroup (runs on across tiles); nxt: tape index of the next chunk to request
    // (wraps at `total`, the chunks one tile consumes: a persistent workgroup streams the tape round and round);
    // b0/b1/b2: LDS buffer (0..2) of chunks cur, cur+1, cur+2
    int cur, nxt, total, tid, lane;
    int b0, b1, b2;
    __device__ __forceinline__ uint4 *buf(int b) const { return lds + b * NLR_CHUNK_SLOTS; }
    // this wave's quarter of chunk c: fragments 8w .. 8w+7, one LDS-DMA instruction each (LDS address = M0 + 16*lane).
    // Inline asm on purpose: behind the builtin hipcc puts an s_waitcnt vmcnt(0) in front of the next ds_read (it cannot
    // tell the DMA's LDS target from the ring reads), which would expose the whole L2 latency once per chunk.  An asm
    // DMA is invisible to hipcc's counters; `landed()` is the one wait that retires it.  M0 is saved and restored.
    __device__ __forceinline__ void dma(int b) {  // tape chunk `nxt` -> LDS buffer b
        const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint64_t g = reinterpret_cast<uint64_t>(base) + (uint64_t)(uint32_t)nxt * (NLR_CHUNK_SLOTS * 16) + w * 8192u;
        const uint32_t l = (uint32_t)(uintptr_t)(nlr_lptr)buf(b) + w * 8192u;
        nxt = (nxt + 1 == total) ? 0 : nxt + 1;
        const uint32_t v = (uint32_t)lane * 16u;
        uint32_t keep;
        // The instruction offset advances the global AND the LDS address (LDS address = M0 + offset + 16 * lane), so four
        // pieces share one M0 value and one SGPR base: 2 x (M0 write + 4 DMA instructions) instead of 8 x (M0 write,
        // 64-bit scalar add, DMA) - the scalar bookkeeping, not the DMA itself, was most of a piece's issue cost.
        asm volatile(
            "s_mov_b32 %[k], m0\n\t"
            "s_mov_b32 m0, %[l]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[v], %[g0]\n\t"
            "global_load_lds_dwordx4 %[v], %[g0] offset:1024\n\t"
            "global_load_lds_dwordx4 %[v], %[g0] offset:2048\n\t"
            "global_load_lds_dwordx4 %[v], %[g0] offset:3072\n\t"
            "s_add_u32 m0, %[l], 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[v], %[g4]\n\t"
            "global_load_lds_dwordx4 %[v], %[g4] offset:1024\n\t"
            "global_load_lds_dwordx4 %[v], %[g4] offset:2048\n\t"
            "global_load_lds_dwordx4 %[v], %[g4] offset:3072\n\t"
            "s_mov_b32 m0, %[k]"
            : [k] "=&s"(keep)
            : [v] "v"(v), [l] "s"(l), [g0] "s"(g), [g4] "s"(g + 4096)
            : "memory", "scc");
    }
    // Workgroup hand-shake without s_barrier.  A barrier per chunk cost 16 % of the kernel (stamps): the four
    // waves drift by a few hundred cycles per chunk and a barrier makes every wave pay the maximum each time.  Instead:
    //   (NLR_SIG_F, c)   own quarter of chunk c+1 has landed (vmcnt) and the last fragment of chunk c-1 is consumed
    //                    -> lane 0 adds 1 to an LDS counter
    //   (NLR_POLL_F-2, c) the counter is read; (NLR_POLL_F, c) spin until it shows 4 (c+1): every wave is past its signal of
    //                    chunk c, so chunk c+1 is complete (read from f = 24 on) and the buffer of chunk c-1 is free -> DMA c+2
    // Waves may now drift by NLR_POLL_F - NLR_SIG_F fragment steps before anyone waits.  LDS accesses of one CU are
    // served in order by one unit, so counter and data need no fence beyond the vmcnt wait in front of the signal.
    // All three steps are single asm statements: any C++ control flow here splits the unrolled MFMA chain into basic
    // blocks and the ds_read / MFMA interleave is lost.  hipcc does not count asm LDS operations; its own counted
    // lgkmcnt waits only get more conservative by that (the counter retires in order).
    uint32_t *sig;
    uint32_t sig_addr, seen;
    __device__ __forceinline__ void signal() {
        uint64_t sv;
        asm volatile(
            "s_waitcnt vmcnt(0)\n\t"
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b64 exec, 1\n\t"
            "ds_add_u32 %[a], %[one]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [sv] "=&s"(sv)
            : [a] "v"(sig_addr), [one] "v"(1u)
            : "memory");
    }
    // The counter is read by an ordinary (volatile, LDS address space) load: hipcc tracks it and puts the exact counted
    // lgkmcnt in front of await()'s asm, whatever it did with the ring reads in between (in padding steps they are dead
    // code, so a hand-counted wait would be wrong there).  Only the rare re-poll inside the spin drains the counter.
    __device__ __forceinline__ void peek() { seen = *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t *>(sig_addr); }
    __device__ __forceinline__ void await() {
        const uint32_t target = 4u * (uint32_t)(cur + 1);
        uint32_t t;
        asm volatile(
            "1:\n\t"
            "v_readfirstlane_b32 %[t], %[seen]\n\t"
            "s_cmp_ge_u32 %[t], %[target]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_sleep 1\n\t"
            "ds_read_b32 %[seen], %[a]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_branch 1b\n"
            "2:"
            : [seen] "+v"(seen), [t] "=&s"(t)
            : [a] "v"(sig_addr), [target] "s"(target)
            : "memory", "scc");
    }
    __device__ __forceinline__ void prologue() {
        cur = 0;
        nxt = 0;
        b0 = 0, b1 = 1, b2 = 2;
        if (tid == 0) *sig = 0u;
        dma(0);
        dma(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // both chunks and the bias block are visible to every wave
#pragma unroll
        for (int f = 0; f < NLR_PF; ++f) ring[f] = buf(0)[f * 64 + lane];
    }
    // bookkeeping at fragment position F of the current chunk; returns the fragment (raw 16 bytes per lane).
    // PAR (parity of the chunk index) is unused by the DMA refill; kept so that the GEMM drivers stay unchanged.
    template <int F, int PAR>
    __device__ __forceinline__ uint4 step() {
        if constexpr (F == NLR_SIG_F) signal();
        if constexpr (F == NLR_POLL_F - 2) peek();
        if constexpr (F == NLR_POLL_F) {
            await();
            dma(b2);
        }
        const uint4 a = ring[F % NLR_PF];
        if constexpr (F + NLR_PF < NLR_CHUNK_FRAGS) ring[F % NLR_PF] = buf(b0)[(F + NLR_PF) * 64 + lane];
        else ring[F % NLR_PF] = buf(b1)[(F + NLR_PF - NLR_CHUNK_FRAGS) * 64 + lane];
        if constexpr (F == NLR_CHUNK_FRAGS - 1) {
            ++cur;
            const int t = b0;
            b0 = b1, b1 = b2, b2 = t;
        }
        return a;
    }
};

template <typename T>
__device__ __forceinline__ T nlr_as(const uint4 &v) {
    return __builtin_bit_cast(T, v);
}

template <int I>
using ic = std::integral_constant<int, I>;

// number of 32-fragment chunks a GEMM occupies on the tape
constexpr int nlr_nch(int ot, int kg, int fps) { return (ot * kg * fps + NLR_CHUNK_FRAGS - 1) / NLR_CHUNK_FRAGS; }

// steps over the unused tail of a GEMM's last chunk (its bookkeeping positions must still run)
template <int F, int PAR>
__device__ __forceinline__ void nlr_pad(Tape &tp) {
    if constexpr (F != 0) {
        (void)tp.template step<F, PAR>();
        nlr_pad<(F + 1) % NLR_CHUNK_FRAGS, PAR>(tp);
    }
}
template <int F, int PAR>
__device__ __forceinline__ void nlr_pad_chunk(Tape &tp) {  // one whole padding chunk
    (void)tp.template step<F, PAR>();
    if constexpr (F + 1 < NLR_CHUNK_FRAGS) nlr_pad_chunk<F + 1, PAR>(tp);
}

// Output-tile-major GEMM driver: for each 32-row output tile o, run all KG k-steps into ONE accumulator, then
// hand the finished tile to `epi`.  Only two accumulator tiles are live (current + the one being post-processed),
// and the epilogue of tile o-1 (ReLU / bf16 conversion / hi-lo split: ~80 VALU instructions) sits in the
// instruction stream right behind the first MFMA of tile o, so it executes in the shadow of tile o's MFMA chain
// instead of serialising between layers (the kernel runs one wave per SIMD: nothing else would hide it).
//   FPS = fragments per k-step (1; 2 for the hi/lo pairs of the split-bf16 path); PAR0 = parity of the first chunk
//   init(ic<o>) -> f32x16 bias tile;  step(acc&, ic<g>, frag0, frag1);  epi(ic<o>, acc)
// epilogue pieces of one finished tile, from piece P0 up to NP (exclusive)
template <int O, int P0, int NP, class Epi>
__device__ __forceinline__ void nlr_epi_rest(const Epi &epi, const f32x16 &acc) {
    if constexpr (P0 < NP) {
        epi(ic<O>{}, ic<P0>{}, acc);
        nlr_epi_rest<O, P0 + 1, NP>(epi, acc);
    }
}
// NPP > 0: `pend(ic<p>)`, p < NPP, are the epilogue pieces of the PREVIOUS GEMM's last output tile; they run in the
//          MFMA shadow of this GEMM's first output tile (whose first k-steps must not read that tile: true for the
//          view layers, where k-step g reads input tile g/2 and the pending tile is the last one).
// DEFER:   the last output tile of this GEMM is handed back raw in `last` (its epilogue becomes the next GEMM's `pend`)
//          instead of being post-processed serially behind the last MFMA, where nothing would hide it.
template <int OT, int KG, int FPS, int NP, int NPP, bool DEFER, int PAR0, int IDX, class Init, class Step, class Epi, class Pend>
__device__ __forceinline__ void nlr_run(Tape &tp, f32x16 &prev, f32x16 &cur, f32x16 &nxt, f32x16 &last, const Init &init,
                                        const Step &step, const Epi &epi, const Pend &pend) {
    // Schedule inside one output tile of KG k-steps (all compile time):
    //   steps D .. D+NP-1 : one piece each of the PREVIOUS tile's epilogue (D = 3 lets that tile's last MFMA retire first)
    //   step  IG          : bias rows of the NEXT tile are read from LDS, a few steps before its first MFMA needs them
    constexpr int D = (KG >= NP + 3) ? 3 : 0;
    constexpr int IG = (KG >= NP + D + 4) ? KG - 4 : KG - 1;
    static_assert(NPP == 0 || KG >= NPP + 3, "pending epilogue needs a long enough first tile");
    if constexpr (IDX < OT * KG) {
        constexpr int o = IDX / KG, g = IDX % KG;
        if constexpr (g == 0) {
            if constexpr (o > 0) {
                prev = cur;
                cur = nxt;
            } else {
                cur = init(ic<0>{});
            }
        }
        constexpr int P0 = (IDX * FPS) % NLR_CHUNK_FRAGS;
        constexpr int PAR = (PAR0 + (IDX * FPS) / NLR_CHUNK_FRAGS) & 1;
        const uint4 f0 = tp.template step<P0, PAR>();
        uint4 f1 = f0;
        if constexpr (FPS == 2) f1 = tp.template step<P0 + 1, PAR>();
        step(cur, ic<g>{}, f0, f1);
        if constexpr (o > 0) {
            if constexpr (g >= D && g - D < NP && g < KG - 1) epi(ic<o - 1>{}, ic<g - D>{}, prev);
            if constexpr (g == KG - 1) nlr_epi_rest<o - 1, (KG - 1 - D < NP ? (KG - 1 - D > 0 ? KG - 1 - D : 0) : NP), NP>(epi, prev);
        } else if constexpr (NPP > 0) {
            if constexpr (g >= 3 && g - 3 < NPP) pend(ic<g - 3>{});
        }
        if constexpr (g == IG && o + 1 < OT) nxt = init(ic<o + 1>{});
        // pin the issue order (fragment read-ahead, MFMA, epilogue piece): left alone, the scheduler sinks the
        // ds_reads next to their use and every MFMA waits out the LDS latency
        __builtin_amdgcn_sched_barrier(0);
        nlr_run<OT, KG, FPS, NP, NPP, DEFER, PAR0, IDX + 1>(tp, prev, cur, nxt, last, init, step, epi, pend);
    } else {
        if constexpr (DEFER) last = cur;
        else nlr_epi_rest<OT - 1, 0, NP>(epi, cur);
        nlr_pad<(OT * KG * FPS) % NLR_CHUNK_FRAGS, (PAR0 + nlr_nch(OT, KG, FPS) - 1) & 1>(tp);
    }
}
// EVEN (unused since the LDS-DMA refill, kept for experiments): one padding chunk after a GEMM with an odd chunk count
// NP: number of epilogue pieces per output tile
template <int OT, int KG, int FPS, int NP, int PAR0, bool EVEN = false, class Init, class Step, class Epi>
__device__ __forceinline__ void nlr_gemm(Tape &tp, const Init &init, const Step &step, const Epi &epi) {
    f32x16 prev, cur, nxt, last;
    nlr_run<OT, KG, FPS, NP, 0, false, PAR0, 0>(tp, prev, cur, nxt, last, init, step, epi, [](auto) {});
    if constexpr (EVEN && (nlr_nch(OT, KG, FPS) & 1)) nlr_pad_chunk<0, (PAR0 + nlr_nch(OT, KG, FPS)) & 1>(tp);
}
// software-pipelined across GEMMs (see nlr_run): pend = previous GEMM's deferred tile, last = this GEMM's
template <int OT, int KG, int NP, int NPP, bool DEFER, int PAR0, bool EVEN = false, class Init, class Step, class Epi, class Pend>
__device__ __forceinline__ void nlr_gemm_pipe(Tape &tp, f32x16 &last, const Init &init, const Step &step, const Epi &epi, const Pend &pend) {
    f32x16 prev, cur, nxt;
    nlr_run<OT, KG, 1, NP, NPP, DEFER, PAR0, 0>(tp, prev, cur, nxt, last, init, step, epi, pend);
    if constexpr (EVEN && (nlr_nch(OT, KG, 1) & 1)) nlr_pad_chunk<0, (PAR0 + nlr_nch(OT, KG, 1)) & 1>(tp);
}

__device__ __forceinline__ f32x16 nlr_bias_tile(const float *bias, int o, int h) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + o * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[q * 4 + e] = v[e];
    }
    return a;
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
// ReLU is applied AFTER the conversion, on the packed pairs: a negative bf16 is a negative int16, so one
// v_pk_max_i16 against zero clears two values (half the VALU work of v_max_f32 per value; -0.0 -> +0.0).
template <bool RELU>
__device__ __forceinline__ void nlr_pack1(TileH &dst, const f32x16 &src) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)src[8 * s + j];
        if (RELU) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            v = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
        }
        dst.f[s] = v;
    }
}
// piece P (0..7) of the pack: values 2P, 2P+1 -> one packed dword of the tile
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool RELU, int P>
__device__ __forceinline__ void nlr_pack_piece(TileH &dst, const f32x16 &src) {
    const f32x2 x = {src[2 * P], src[2 * P + 1]};
    bf16x2 v = __builtin_convertvector(x, bf16x2);  // one v_cvt_pk_bf16_f32 (RNE)
    if (RELU) {
        const s16x2 z = {0, 0};  // max(bf16 bits as i16, 0): relu on the packed pair, -0 and negatives -> +0
        v = __builtin_bit_cast(bf16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
    }
    dst.f[P >> 2][2 * (P & 3)] = v[0];
    dst.f[P >> 2][2 * (P & 3) + 1] = v[1];
}
// hi = bf16(x), lo = bf16(x - hi)   (x - hi is exact in f32)
template <bool RELU>
__device__ __forceinline__ void nlr_split1(TileH &hi, TileH &lo, const f32x16 &src) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = src[8 * s + j];
            if (RELU) v = fmaxf(v, 0.0f);
            const __bf16 hh = (__bf16)v;
            hi.f[s][j] = hh;
            lo.f[s][j] = (__bf16)(v - (float)hh);
        }
}
template <bool RELU, int P>
__device__ __forceinline__ void nlr_split_piece(TileH &hi, TileH &lo, const f32x16 &src) {
    f32x2 x = {src[2 * P], src[2 * P + 1]};
    if (RELU) x = __builtin_elementwise_max(x, (f32x2){0.0f, 0.0f});
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const bf16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), bf16x2);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        hi.f[P >> 2][2 * (P & 3) + e] = h[e];
        lo.f[P >> 2][2 * (P & 3) + e] = l[e];
    }
}
template <bool RELU>
__device__ __forceinline__ f32x16 nlr_act(const f32x16 &src) {
    f32x16 r = src;
    if (RELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = fmaxf(r[i], 0.0f);
    }
    return r;
}

// MFMA steps ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void nlr_mma_bf16(f32x16 &acc, const uint4 &a, const bf16x8 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nlr_as<bf16x8>(a), b, acc, 0, 0, 0);
}
// Split-bf16 ("bf16x3"): W = Wh + Wl, x = xh + xl (each part bf16), W.x ~= Wh.xh + Wh.xl + Wl.xh with f32
// accumulation: 16 mantissa bits per operand (relative error ~2^-16) at 3/16 of the exact-f32 MFMA cost.
__device__ __forceinline__ void nlr_mma_x3(f32x16 &acc, const uint4 &ah, const uint4 &al, const bf16x8 &bh, const bf16x8 &bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nlr_as<bf16x8>(ah), bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nlr_as<bf16x8>(ah), bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nlr_as<bf16x8>(al), bh, acc, 0, 0, 0);
}
// exact-f32 MFMA: one fragment (float4 per lane) carries 4 k-steps of 2 features
template <int G, int KT>
__device__ __forceinline__ void nlr_mma_f32(f32x16 &acc, const uint4 &a, const f32x16 (&in)[KT]) {
    const f32x4 af = nlr_as<f32x4>(a);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], in[G >> 2][(G & 3) * 4 + e], acc, 0, 0, 0);
}

// WT = view width / 32, BT = bottleneck / 32, FG = ceil(F/8), HT = head hidden tiles (0, 2 or 4)
// PREC: NLR_PREC_F32 (all f32), NLR_PREC_MIXED (trunk+heads f32, view bf16), NLR_PREC_FAST (trunk+heads bf16x3, view bf16)
#define NLR_BIAS_MAX 4096  // floats of LDS reserved for the bias block (16 KiB)
template <int WT, int BT, int FG, int HT, int PREC>
__global__ void __launch_bounds__(256, 1) nlr_mlp_kernel(MlpParams P) {
    __shared__ __align__(16) uint4 lds_tape[NLR_NBUF * NLR_CHUNK_SLOTS];
    __shared__ __align__(16) float lds_bias[NLR_BIAS_MAX];
    __shared__ uint32_t lds_sig;
    constexpr bool VIEW_F32 = (PREC == NLR_PREC_F32);
    constexpr bool X3 = (PREC == NLR_PREC_FAST);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    constexpr int FT = (FG + 3) / 4;
    constexpr int HTA = HT > 0 ? HT : 1;
    // bias block offsets (floats)
    constexpr int OB_D0 = 0, OB_D2 = 64, OB_H1 = OB_D2 + BT * 32, OB_H2 = OB_H1 + HT * 32, OB_V0 = OB_H2 + 32;
    constexpr int OB_V1 = OB_V0 + WT * 32, OB_VL = OB_V1 + WT * 32;
    // chunk parity at the start of every GEMM of the fixed sequence (see Tape::step)
    constexpr int CF = X3 ? 2 : 1;                     // fragments per k-step in the trunk/heads
    constexpr int KU = X3 ? 2 : 4;                     // k-steps per 32-feature input tile in the trunk/heads
    constexpr int KV = VIEW_F32 ? 4 : 2;               // ... in the view MLP
    constexpr int KD0 = X3 ? (FG + 1) / 2 : FG;
    constexpr int P_D0 = 0;
    constexpr int P_D2 = P_D0 + nlr_nch(2, KD0, CF);
    constexpr int P_H1 = P_D2 + nlr_nch(BT, 2 * KU, CF);
    constexpr int P_H2 = P_H1 + (HT > 0 ? nlr_nch(HT, BT * KU, CF) : 0);
    constexpr int P_V0 = P_H2 + (HT > 0 ? nlr_nch(1, HT * KU, CF) : 0);
    constexpr int P_V1 = P_V0 + nlr_nch(WT, (BT + 1) * KV, 1);
    constexpr int P_VL = P_V1 + nlr_nch(WT, (WT + BT + 1) * KV, 1);

    // ---- persistent workgroup: one per CU, tiles (4 waves x 32 samples) taken round-robin.  The bias block is staged
    // once, the weight tape streams round and round (its read-ahead runs across the tile seam into chunk 0 of the next
    // tile), and the next tile's inputs are requested as soon as this tile's are unpacked: no prologue, dispatch gap or
    // exposed load latency between tiles.
    const uint32_t ntiles = (P.M + 127) / 128;
    auto load_inputs = [&](uint32_t tile, f32x4 (&fv)[FG], f32x4 (&ev)[4]) {
        const uint32_t smp = (tile * 4 + wave) * 32 + col;
        const uint32_t sc = smp < P.M ? smp : P.M - 1;
        const float *fp = P.feat + (size_t)sc * P.F;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            const uint32_t f0 = 8 * g + 4 * h;
            fv[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (f0 + 4 <= P.F)
                fv[g] = P.feat_piece_major ? *reinterpret_cast<const f32x4 *>(P.feat + ((size_t)(f0 >> 2) * P.M + sc) * 4)
                                           : *reinterpret_cast<const f32x4 *>(fp + f0);
        }
        const uint32_t ray = sc / P.S;
#pragma unroll
        for (int q = 0; q < 4; ++q) ev[q] = *reinterpret_cast<const f32x4 *>(P.enc + (size_t)ray * 32 + 8 * q + 4 * h);
    };
    f32x4 fv[FG], ev[4];
    load_inputs(blockIdx.x, fv, ev);
    for (uint32_t i = threadIdx.x * 4; i < P.bias_count; i += 1024)
        *reinterpret_cast<f32x4 *>(lds_bias + i) = *reinterpret_cast<const f32x4 *>(P.bias_all + i);

    Tape tp;
    tp.base = P.tape;
    tp.lds = lds_tape;
    tp.sig = &lds_sig;
    tp.sig_addr = (uint32_t)(uintptr_t)(nlr_lptr)&lds_sig;
    tp.total = P.rgb ? (int)P.tape_chunks : P_V0;  // without the view MLP a tile consumes the trunk + head chunks only
    tp.tid = threadIdx.x;
    tp.lane = lane;
    tp.prologue();  // ends with __syncthreads(): the bias block is visible too

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t sample = (tile * 4 + wave) * 32 + col;
    const bool valid = sample < P.M;
    // ---- features / direction encoding -> accumulator-layout tiles (lane half h holds rows 8q+4h..+3 of each group)
    f32x16 fin[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[t][r] = 0.0f;
#pragma unroll
    for (int g = 0; g < FG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[g >> 2][(g & 3) * 4 + e] = fv[g][e];
    f32x16 encf;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) encf[q * 4 + e] = ev[q][e];
    if (tile + gridDim.x < ntiles) load_inputs(tile + gridDim.x, fv, ev);

    float raw_density = 0.0f;
    f32x16 lo;  // [K logits | intensity] output tile of the heads
#pragma unroll
    for (int r = 0; r < 16; ++r) lo[r] = 0.0f;
    TileH hbe[VIEW_F32 ? 1 : BT + 1];      // bf16 [bottleneck | dir-enc] tiles for the view MLP
    f32x16 hbf[VIEW_F32 ? BT + 1 : 1];     // the same in f32 (NLR_PREC_F32)

    if constexpr (X3) {
        // ---- density trunk + heads on split-bf16
        constexpr int FK = (FG + 1) / 2;  // 16-feature k-steps covering the grid features
        TileH fh[FT], fl[FT];
#pragma unroll
        for (int t = 0; t < FT; ++t) nlr_split1<false>(fh[t], fl[t], fin[t]);
        TileH dh[2], dl[2];
        nlr_gemm<2, FK, 2, 8, P_D0 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_D0, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &f1) {
                constexpr int G = decltype(g)::value;
                nlr_mma_x3(a, f0, f1, fh[G >> 1].f[G & 1], fl[G >> 1].f[G & 1]);
            },
            [&](auto o, auto p, const f32x16 &a) { nlr_split_piece<true, decltype(p)::value>(dh[decltype(o)::value], dl[decltype(o)::value], a); });
        TileH hbl[BT];
        nlr_gemm<BT, 4, 2, 8, P_D2 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_D2, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &f1) {
                constexpr int G = decltype(g)::value;
                nlr_mma_x3(a, f0, f1, dh[G >> 1].f[G & 1], dl[G >> 1].f[G & 1]);
            },
            [&](auto o, auto p, const f32x16 &a) {
                constexpr int O = decltype(o)::value, Pc = decltype(p)::value;
                if constexpr (O == 0 && Pc == 0) raw_density = a[0];
                nlr_split_piece<false, Pc>(hbe[O], hbl[O], a);
            });
        if constexpr (HT > 0) {
            TileH qh[HTA], ql[HTA];
            nlr_gemm<HT, BT * 2, 2, 8, P_H1 & 1>(
                tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_H1, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &f1) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_x3(a, f0, f1, hbe[G >> 1].f[G & 1], hbl[G >> 1].f[G & 1]);
                },
                [&](auto o, auto p, const f32x16 &a) { nlr_split_piece<true, decltype(p)::value>(qh[decltype(o)::value], ql[decltype(o)::value], a); });
            nlr_gemm<1, HT * 2, 2, 1, P_H2 & 1>(
                tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_H2, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &f1) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_x3(a, f0, f1, qh[G >> 1].f[G & 1], ql[G >> 1].f[G & 1]);
                },
                [&](auto, auto, const f32x16 &a) { lo = a; });
        }
    } else {
        // ---- density trunk + heads on the exact-f32 MFMA
        f32x16 hid[2];
        nlr_gemm<2, FG, 1, 1, P_D0 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_D0, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, FT>(a, f0, fin); },
            [&](auto o, auto, const f32x16 &a) { hid[decltype(o)::value] = nlr_act<true>(a); });
        f32x16 hb[BT];
        nlr_gemm<BT, 8, 1, 1, P_D2 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_D2, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, 2>(a, f0, hid); },
            [&](auto o, auto, const f32x16 &a) {
                constexpr int O = decltype(o)::value;
                if constexpr (O == 0) raw_density = a[0];
                hb[O] = a;
                if constexpr (VIEW_F32) hbf[O] = a; else nlr_pack1<false>(hbe[O], a);
            });
        if constexpr (HT > 0) {
            f32x16 hh[HTA];
            nlr_gemm<HT, BT * 4, 1, 1, P_H1 & 1>(
                tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_H1, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, BT>(a, f0, hb); },
                [&](auto o, auto, const f32x16 &a) { hh[decltype(o)::value] = nlr_act<true>(a); });
            nlr_gemm<1, HT * 4, 1, 1, P_H2 & 1>(
                tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_H2, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, HTA>(a, f0, hh); },
                [&](auto, auto, const f32x16 &a) { lo = a; });
        }
    }
    if (h == 0 && valid) {
        const float x = raw_density + P.density_bias;
        P.density[sample] = x > 20.0f ? x : log1pf(expf(x));
    }
    // ---- semantic / intensity outputs: rows [0,K) logits -> softmax, row int_row -> intensity.
    // Per-sample heads are stored class-major ([K, M], [3, M]): one store instruction writes two 128-byte runs.
    if constexpr (HT > 0) {
        if (P.K > 0) {  // softmax over rows [0,K) of this column, split over the two lane halves
            // Branch-free: rows >= K take -inf (exp -> 0), so the only exec-masked instructions are the stores; a
            // per-row `if (row < K)` differs between the lane halves and costs an exec save/restore per row.
            float e[16], mx = -INFINITY, s = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                e[r] = nlr_row(r, h) < (int)P.K ? lo[r] : -INFINITY;
                mx = fmaxf(mx, e[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                e[r] = expf(e[r] - mx);
                s += e[r];
            }
            s += __shfl_xor(s, 32, 64);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (valid && nlr_row(r, h) < (int)P.K) P.sem[(size_t)nlr_row(r, h) * P.M + sample] = e[r] / s;
        }
        if (P.inten && valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nlr_row(r, h) == (int)P.int_row) P.inten[sample] = lo[r];
        }
    }
    if (P.rgb == nullptr) continue;  // density/semantic/intensity only (uniform for the whole grid)

    // ---- view MLP.  Layer 0 input = [bottleneck | enc]; layer 1 input = [x | bottleneck | enc] (skip concat,
    // models.py:1227-1228); the 27 dir-encoding features ride as one extra zero-padded 32-feature input tile.
    f32x16 out1;
    if constexpr (!VIEW_F32) {
        nlr_pack1<false>(hbe[BT], encf);
        TileH x[WT], y[WT];
        // The last output tile of every layer is carried raw (`cx` / `cy`) and packed in the MFMA shadow of the next
        // layer's first tile: x/y[WT-1] is read last there (k-steps 2 WT - 2, 2 WT - 1), so nothing waits for it.
        // Needs 2 WT - 2 > 6 k-steps ahead of the first read of that tile: WT >= 6; narrower layers run the serial form.
        constexpr bool PIPE = WT >= 6;
        constexpr int NPP = PIPE ? 8 : 0;
        f32x16 cx, cy;
        nlr_gemm_pipe<WT, (BT + 1) * 2, 8, 0, PIPE, P_V0 & 1>(
            tp, cx, [&](auto o) { return nlr_bias_tile(lds_bias + OB_V0, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                constexpr int G = decltype(g)::value;
                nlr_mma_bf16(a, f0, hbe[G >> 1].f[G & 1]);
            },
            [&](auto o, auto p, const f32x16 &a) { nlr_pack_piece<true, decltype(p)::value>(x[decltype(o)::value], a); }, [](auto) {});
        nlr_gemm_pipe<WT, (WT + BT + 1) * 2, 8, NPP, PIPE, P_V1 & 1>(
            tp, cy, [&](auto o) { return nlr_bias_tile(lds_bias + OB_V1, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                constexpr int G = decltype(g)::value;
                if constexpr (G < 2 * WT) nlr_mma_bf16(a, f0, x[G >> 1].f[G & 1]);
                else nlr_mma_bf16(a, f0, hbe[(G - 2 * WT) >> 1].f[G & 1]);
            },
            [&](auto o, auto p, const f32x16 &a) { nlr_pack_piece<true, decltype(p)::value>(y[decltype(o)::value], a); },
            [&](auto p) { nlr_pack_piece<true, decltype(p)::value>(x[WT - 1], cx); });
        // hidden layers 2..D-1, two per iteration (y -> x -> y) so that no tile copies are needed
        uint32_t l = 2;
        for (; l + 1 < P.depth; l += 2) {
            const float *bl = lds_bias + OB_VL + (l - 2) * (WT * 32);
            nlr_gemm_pipe<WT, WT * 2, 8, NPP, PIPE, P_VL & 1>(
                tp, cx, [&](auto o) { return nlr_bias_tile(bl, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_bf16(a, f0, y[G >> 1].f[G & 1]);
                },
                [&](auto o, auto p, const f32x16 &a) { nlr_pack_piece<true, decltype(p)::value>(x[decltype(o)::value], a); },
                [&](auto p) { nlr_pack_piece<true, decltype(p)::value>(y[WT - 1], cy); });
            nlr_gemm_pipe<WT, WT * 2, 8, NPP, PIPE, P_VL & 1>(
                tp, cy, [&](auto o) { return nlr_bias_tile(bl + WT * 32, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_bf16(a, f0, x[G >> 1].f[G & 1]);
                },
                [&](auto o, auto p, const f32x16 &a) { nlr_pack_piece<true, decltype(p)::value>(y[decltype(o)::value], a); },
                [&](auto p) { nlr_pack_piece<true, decltype(p)::value>(x[WT - 1], cx); });
        }
        if (l < P.depth) {  // odd number of hidden layers: one more, result moved back into y
            const float *bl = lds_bias + OB_VL + (l - 2) * (WT * 32);
            nlr_gemm_pipe<WT, WT * 2, 8, NPP, false, P_VL & 1>(
                tp, cx, [&](auto o) { return nlr_bias_tile(bl, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_bf16(a, f0, y[G >> 1].f[G & 1]);
                },
                [&](auto o, auto p, const f32x16 &a) { nlr_pack_piece<true, decltype(p)::value>(x[decltype(o)::value], a); },
                [&](auto p) { nlr_pack_piece<true, decltype(p)::value>(y[WT - 1], cy); });
#pragma unroll
            for (int t = 0; t < WT; ++t) y[t] = x[t];
            cy = cx;  // not pending any more: y[WT-1] is complete (the flag below is compile-time, see `odd`)
        }
        const bool odd = ((P.depth - 2) & 1) != 0;
        if (odd) {
            nlr_gemm_pipe<1, WT * 2, 1, 0, false, P_VL & 1>(
                tp, cx, [&](auto o) { return nlr_bias_tile(lds_bias + OB_VL + (P.depth - 2) * (WT * 32), o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_bf16(a, f0, y[G >> 1].f[G & 1]);
                },
                [&](auto, auto, const f32x16 &a) { out1 = a; }, [](auto) {});
        } else {
            nlr_gemm_pipe<1, WT * 2, 1, NPP, false, P_VL & 1>(
                tp, cx, [&](auto o) { return nlr_bias_tile(lds_bias + OB_VL + (P.depth - 2) * (WT * 32), o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                    constexpr int G = decltype(g)::value;
                    nlr_mma_bf16(a, f0, y[G >> 1].f[G & 1]);
                },
                [&](auto, auto, const f32x16 &a) { out1 = a; },
                [&](auto p) { nlr_pack_piece<true, decltype(p)::value>(y[WT - 1], cy); });
        }
    } else {
        hbf[BT] = encf;
        f32x16 x[WT], y[WT];
        nlr_gemm<WT, (BT + 1) * 4, 1, 1, P_V0 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_V0, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, BT + 1>(a, f0, hbf); },
            [&](auto o, auto, const f32x16 &a) { x[decltype(o)::value] = nlr_act<true>(a); });
        nlr_gemm<WT, (WT + BT + 1) * 4, 1, 1, P_V1 & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_V1, o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) {
                constexpr int G = decltype(g)::value;
                if constexpr (G < 4 * WT) nlr_mma_f32<G, WT>(a, f0, x);
                else nlr_mma_f32<G - 4 * WT, BT + 1>(a, f0, hbf);
            },
            [&](auto o, auto, const f32x16 &a) { y[decltype(o)::value] = nlr_act<true>(a); });
        for (uint32_t l = 2; l < P.depth; ++l) {
            const float *bl = lds_bias + OB_VL + (l - 2) * (WT * 32);
            nlr_gemm<WT, WT * 4, 1, 1, P_VL & 1>(
                tp, [&](auto o) { return nlr_bias_tile(bl, o.value, h); },
                [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, WT>(a, f0, y); },
                [&](auto o, auto, const f32x16 &a) { x[decltype(o)::value] = nlr_act<true>(a); });
#pragma unroll
            for (int t = 0; t < WT; ++t) y[t] = x[t];
        }
        nlr_gemm<1, WT * 4, 1, 1, P_VL & 1>(
            tp, [&](auto o) { return nlr_bias_tile(lds_bias + OB_VL + (P.depth - 2) * (WT * 32), o.value, h); },
            [&](f32x16 &a, auto g, const uint4 &f0, const uint4 &) { nlr_mma_f32<decltype(g)::value, WT>(a, f0, y); },
            [&](auto, auto, const f32x16 &a) { out1 = a; });
    }
    if (h == 0 && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float z = P.rgb_premul * out1[c] + P.rgb_bias;
            const float sg = 1.0f / (1.0f + expf(-z));
            P.rgb[(size_t)c * P.M + sample] = sg * (1.0f + 2.0f * P.rgb_padding) - P.rgb_padding;
        }
    }
    }  // tile loop
    // the read-ahead DMA must not outlive the workgroup's LDS allocation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// One explicit instance per translation unit (nlr_mlp_inst.hip is compiled once per (WT, HT, PREC) by the Makefile:
// the fully unrolled GEMM chain is slow to compile, so the instances build in parallel).
#define NLR_MLP_LAUNCH_NAME2(wt, ht, pr) nlr_mlp_launch_##wt##_##ht##_##pr
#define NLR_MLP_LAUNCH_NAME(wt, ht, pr) NLR_MLP_LAUNCH_NAME2(wt, ht, pr)
#define NLR_MLP_DECLARE(wt, ht, pr) void NLR_MLP_LAUNCH_NAME(wt, ht, pr)(const MlpParams &P, dim3 grid, hipStream_t st)
