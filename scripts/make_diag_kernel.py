"""Diagnostic only (never shipped): derive a stamp-instrumented copy of csrc/nlr_mlp_kernel.h.

    python scripts/make_diag_kernel.py OUT_DIR
writes OUT_DIR/diag_kernel.h + OUT_DIR/diag_inst.hip.  Build one instance per ablation mask with -DNLR_DIAG=<mask>
and link it in place of build/nlr_mlp_inst_8_4_2.o (scripts/diag_build.sh); scripts/stamp_probe.py reads the stamps.
  s_memtime stamps at the phase boundaries of the FAST (bf16 view MLP) path go behind the intensity buffer.
  NLR_DIAG bits: 1 no tape refill (DMA), 2 epilogue reduced to register moves, 4 no fragment ring reads,
                 8 no per-chunk hand-shake (signal/peek/await).   Masked builds compute garbage: timing only.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "nerf-lidar_amd", "csrc", "nlr_mlp_kernel.h")).read()


def sub(s, old, new, count=1):
    assert old in s, old[:80]
    return s.replace(old, new, count)


s = src
s = sub(s, '#include "nlr_kernels.h"', '#include "%s/nerf-lidar_amd/csrc/nlr_kernels.h"\n#ifndef NLR_DIAG\n#define NLR_DIAG 0\n#endif' % ROOT)
# ablation bits
s = sub(s, "        if constexpr (F == NLR_SIG_F) signal();\n        if constexpr (F == NLR_POLL_F - 2) peek();\n        if constexpr (F == NLR_POLL_F) {\n            await();\n            dma(b2);\n        }",
        "#if !(NLR_DIAG & 8)\n        if constexpr (F == NLR_SIG_F) signal();\n        if constexpr (F == NLR_POLL_F - 2) peek();\n        if constexpr (F == NLR_POLL_F) await();\n#endif\n#if !(NLR_DIAG & 1)\n        if constexpr (F == NLR_POLL_F) dma(b2);\n#endif")
s = sub(s, "        if constexpr (F + NLR_PF < NLR_CHUNK_FRAGS) ring[F % NLR_PF] = buf(b0)[(F + NLR_PF) * 64 + lane];\n"
           "        else ring[F % NLR_PF] = buf(b1)[(F + NLR_PF - NLR_CHUNK_FRAGS) * 64 + lane];",
        "#if !(NLR_DIAG & 4)\n        if constexpr (F + NLR_PF < NLR_CHUNK_FRAGS) ring[F % NLR_PF] = buf(b0)[(F + NLR_PF) * 64 + lane];\n"
        "        else ring[F % NLR_PF] = buf(b1)[(F + NLR_PF - NLR_CHUNK_FRAGS) * 64 + lane];\n#endif")
s = sub(s, "    const f32x2 x = {src[2 * P], src[2 * P + 1]};\n    bf16x2 v = __builtin_convertvector(x, bf16x2);",
        "#if NLR_DIAG & 2\n    dst.f[P >> 2][2 * (P & 3)] = __builtin_bit_cast(bf16x2, src[2 * P])[1];\n"
        "    dst.f[P >> 2][2 * (P & 3) + 1] = __builtin_bit_cast(bf16x2, src[2 * P])[0];\n    return;\n#endif\n"
        "    const f32x2 x = {src[2 * P], src[2 * P + 1]};\n    bf16x2 v = __builtin_convertvector(x, bf16x2);")
# stamps
s = sub(s, "    Tape tp;\n",
        "    unsigned long long stamps[12]; int ns = 0;\n"
        "#define STAMP() do { __builtin_amdgcn_sched_barrier(0); stamps[ns++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)\n"
        "    Tape tp;\n")
s = sub(s, "    const bool valid = sample < P.M;\n",
        "    const bool valid = sample < P.M;\n    ns = 0;\n    STAMP();\n    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();\n")
s = sub(s, "    if (h == 0 && valid) {\n        const float x = raw_density + P.density_bias;", "    STAMP();  // 1: trunk + heads done\n    if (h == 0 && valid) {\n        const float x = raw_density + P.density_bias;")
s = sub(s, "    // ---- view MLP.", "    STAMP();  // 2: head outputs stored\n    // ---- view MLP.")
s = sub(s, "        nlr_gemm_pipe<WT, (WT + BT + 1) * 2, 8, NPP, PIPE, P_V1 & 1>(", "        STAMP();  // 3: V0 done\n        nlr_gemm_pipe<WT, (WT + BT + 1) * 2, 8, NPP, PIPE, P_V1 & 1>(")
s = sub(s, "        // hidden layers 2..D-1, two per iteration", "        STAMP();  // 4: V1 done\n        // hidden layers 2..D-1, two per iteration")
s = sub(s, "        if (l < P.depth) {  // odd number of hidden layers", "        STAMP();  // 5: hidden pairs done\n        if (l < P.depth) {  // odd number of hidden layers")
s = sub(s, "    } else {\n        hbf[BT] = encf;", "        STAMP();  // 6: rgb layer done\n    } else {\n        hbf[BT] = encf;")
s = sub(s, "    if (h == 0 && valid) {\n#pragma unroll\n        for (int c = 0; c < 3; ++c) {\n            const float z = P.rgb_premul",
        "    if (threadIdx.x == 0 && blockIdx.x < 4096 && P.inten && tile + gridDim.x >= ntiles) {\n"
        "        unsigned long long *dbg = (unsigned long long *)(P.inten + P.M) + (size_t)blockIdx.x * 16;\n"
        "        for (int i = 0; i < 7; ++i) dbg[i] = stamps[i];\n"
        "        dbg[7] = __builtin_amdgcn_s_memtime();\n        dbg[8] = rt0;\n        dbg[9] = __builtin_amdgcn_s_memrealtime();\n    }\n"
        "    if (h == 0 && valid) {\n#pragma unroll\n        for (int c = 0; c < 3; ++c) {\n            const float z = P.rgb_premul")

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "diag_kernel.h"), "w").write(s)
open(os.path.join(out, "diag_inst.hip"), "w").write(
    '#include "diag_kernel.h"\nNLR_MLP_DECLARE(NLR_INST_WT, NLR_INST_HT, NLR_INST_PREC) {\n'
    "    hipLaunchKernelGGL((nlr_mlp_kernel<NLR_INST_WT, 8, 5, NLR_INST_HT, NLR_INST_PREC>), grid, dim3(256), 0, st, P);\n}\n")
print("stamps:", s.count("STAMP();") - 0)
