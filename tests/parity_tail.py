#!/usr/bin/env python3
"""Where does the full-size parity tail come from?  ("What's weak" 1.)

Renders the benchmarked sweep (C2, full-size fp32 tables, 32 x 1024 rays) at the three precisions with the per-sample history,
runs the CPU oracle (fp32) on `--rays` rays of it, and reports

  1. outlier statistics per precision (depth L1 / p95 / max, fractions of rays with |d depth| > 1e-3 and > 1e-2,
     |d intensity| > 1e-3, label mismatches);
  2. a float64 evaluation of the SAME algorithm (the oracle's functions on float64 tensors, the grid interpolation in
     float64) on a subset of rays: how far is the *reference's own fp32 arithmetic* from the exact result of its algorithm on
     those rays, next to how far the GPU is from it;
  3. stage isolation on the worst rays: every stage of every level (resample, ray warp, cast+encode+MLP, alpha weights) is
     re-evaluated by the oracle ON THE GPU'S OWN INPUTS of that stage, so a stage's error is seen before the next stages
     amplify it; plus the first level / fencepost at which the GPU's sdist leaves the oracle's.

Test infrastructure (imports oracle/); writes text to stdout.  Run on the GPU box:
    python tests/parity_tail.py --rays 8192 > gpurun_out/parity_tail.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
from oracle import nlr_oracle as orc

PRECS = [(_lib.PREC_F32, "F32"), (_lib.PREC_MIXED, "MIXED"), (_lib.PREC_FAST, "FAST")]


def ulps(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp


class GridEncoder64(orc.GridEncoder):
    """The hash-grid interpolation of oracle/grid_oracle.c in float64 positions / weights / sums (same f32 level scales,
    same indices unless a position sits within 1e-7 of a cell face)."""

    def __call__(self, inputs, bound=1):
        x01 = ((inputs + bound) / (2 * bound)).reshape(-1, 3).numpy().astype(np.float64)
        B = x01.shape[0]
        L, C = self.num_levels, self.level_dim
        out = np.zeros((B, L, C), np.float64)
        sc, rs = orc.level_scale(L, self.S, self.H)
        oob = ((x01 < 0) | (x01 > 1)).any(-1)
        table = self.table.astype(np.float64)
        P = np.array([1, 2654435761, 805459861], np.uint64)
        for l in range(L):
            hs = int(self.offsets[l + 1] - self.offsets[l])
            step = int(rs[l]) + 1
            pos = x01 * float(sc[l]) + 0.5
            pg = np.floor(pos)
            fr = pos - pg
            pg = pg.astype(np.int64)
            dense = step ** 3 <= hs
            for c8 in range(8):
                w = np.ones(B)
                pl = np.empty((B, 3), np.int64)
                for d in range(3):
                    bit = (c8 >> d) & 1
                    w = w * (fr[:, d] if bit else 1 - fr[:, d])
                    pl[:, d] = pg[:, d] + bit
                if dense:
                    idx = pl[:, 0] + pl[:, 1] * step + pl[:, 2] * step * step
                else:
                    u = pl.astype(np.uint64)
                    idx = ((u[:, 0] * P[0]) ^ ((u[:, 1] * P[1]) & np.uint64(0xFFFFFFFF)) ^ ((u[:, 2] * P[2]) & np.uint64(0xFFFFFFFF)))
                    idx = (idx & np.uint64(0xFFFFFFFF)).astype(np.int64)
                idx = np.where(oob, 0, idx % hs) + int(self.offsets[l])
                out[:, l] += w[:, None] * table[idx]
            out[oob, l] = 0
        return torch.from_numpy(out.reshape(list(inputs.shape[:-1]) + [L * C]))


def oracle_run(sd, mc, batch_np, rows, dtype=torch.float32, chunk=1024, enc=None):
    sdt = {k: v.to(dtype) for k, v in orc.to_torch_sd(sd).items()}
    enc = enc or orc.make_encoders(sd, mc)
    b = {k: torch.from_numpy(np.ascontiguousarray(v[rows])).to(dtype) for k, v in batch_np.items()}
    rend, hist = [], []
    with torch.no_grad():
        for i in range(0, len(rows), chunk):
            r, h = orc.model_forward(sd, mc, {k: v[i:i + chunk] for k, v in b.items()}, encoders=enc, sd_t=sdt)
            rend.append({k: v for k, v in r[-1].items()})
            hist.append([{k: lv[k] for k in ("sdist", "tdist", "weights", "density")} for lv in h])
    R = {k: torch.cat([x[k] for x in rend]).numpy() for k in rend[0]}
    H = [{k: torch.cat([c[l][k] for c in hist]).numpy() for k in hist[0][l]} for l in range(len(hist[0]))]
    return R, H


def stats(name, g, ref, out):
    d = np.abs(g["depth"] - ref["depth"])
    i = np.abs(g["intensity"] - ref["intensity"])
    lab = g["semantic"].argmax(-1) != ref["semantic"].argmax(-1)
    out.append(f"{name:34s} depth L1 {d.mean():.3e} p95 {np.percentile(d, 95):.2e} p99 {np.percentile(d, 99):.2e} max {d.max():.3e} | "
               f"frac>1e-3 {np.mean(d > 1e-3):.5f} ({int((d > 1e-3).sum())}) frac>1e-2 {np.mean(d > 1e-2):.5f} ({int((d > 1e-2).sum())}) | "
               f"intensity L1 {i.mean():.2e} max {i.max():.2e} frac>1e-3 {np.mean(i > 1e-3):.5f} ({int((i > 1e-3).sum())}) | "
               f"label mismatches {int(lab.sum())}")
    return d, i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--rays64", type=int, default=768)
    ap.add_argument("--worst", type=int, default=8)
    ap.add_argument("--log2-hashmap", type=int, default=None)
    ap.add_argument("--width", type=int, default=1024)
    args = ap.parse_args()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    out = []
    mc = nconfig.workload("C2", args.log2_hashmap)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    full = nlidar.synthetic_sweep(width=args.width, seed=0)
    n = full["origins"].shape[0]
    idx = np.linspace(0, n - 1, args.rays).astype(np.int64)
    samples = mc.level_samples()
    t0 = time.time()
    ref, refh = oracle_run(sd, mc, full, idx)
    out.append(f"# oracle fp32 on {len(idx)} rays: {time.time() - t0:.1f} s")

    dev = torch.device("cuda:0")
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in full.items()}
    G, GH = {}, {}
    for prec, pname in PRECS:
        model = Model(mc, sd, device=dev, precision=prec)
        r, h = model.render_rays(batch, compute_extras=True, scale_factor=1 / 250, want_history=True)
        torch.cuda.synchronize()
        G[pname] = {k: v.cpu().numpy()[idx] for k, v in r.items() if k in ("depth", "intensity", "semantic", "labels")}
        GH[pname] = [{k: lv[k].cpu().numpy()[idx] for k in ("sdist", "tdist", "weights", "density")} for lv in h]
        # the render path (compositing mode, no history) is what bench.py times
        r2, _ = model.render_rays(batch, compute_extras=True, scale_factor=1 / 250, want_history=False)
        torch.cuda.synchronize()
        G[pname + "/render"] = {k: v.cpu().numpy()[idx] for k, v in r2.items() if k in ("depth", "intensity", "semantic", "labels")}
        del model
    out.append("\n## 1. GPU against the fp32 oracle, %d rays of the benchmarked sweep" % len(idx))
    D = {}
    for name in G:
        D[name], _ = stats("GPU " + name + " vs oracle32", G[name], ref, out)
    out.append("\n   GPU precisions against each other (same rays):")
    stats("GPU FAST vs GPU F32", G["FAST"], G["F32"], out)
    stats("GPU MIXED vs GPU F32", G["MIXED"], G["F32"], out)
    stats("GPU FAST/render vs GPU FAST", G["FAST/render"], G["FAST"], out)

    # ---- 2. float64 truth on a subset: the worst rays of every precision + an even sample -------------------------------
    worst = np.unique(np.concatenate([np.argsort(-D[p])[:args.rays64 // 6] for p in ("F32", "MIXED", "FAST")]))
    even = np.linspace(0, len(idx) - 1, args.rays64 - len(worst)).astype(np.int64)
    sub = np.unique(np.concatenate([worst, even]))
    torch.set_default_dtype(torch.float64)
    t0 = time.time()
    enc64 = {k: GridEncoder64(e.table, e.offsets, e.grid_sizes.numpy(), 2.0 ** e.S, e.H) for k, e in orc.make_encoders(sd, mc).items()}
    ref64, refh64 = oracle_run(sd, mc, full, idx[sub], dtype=torch.float64, chunk=256, enc=enc64)
    torch.set_default_dtype(torch.float32)
    out.append(f"\n## 2. float64 evaluation of the same algorithm on {len(sub)} rays ({len(worst)} = the worst rays of the three precisions, "
               f"the rest evenly spaced): {time.time() - t0:.1f} s")
    pick = lambda d: {k: v[sub] for k, v in d.items()}
    stats("oracle32 vs float64", pick(ref), ref64, out)
    for name in ("F32", "MIXED", "FAST", "FAST/render"):
        stats("GPU " + name + " vs float64", pick(G[name]), ref64, out)
    ev = np.isin(sub, even) & ~np.isin(sub, worst)
    out.append("   evenly spaced rays only (%d):" % int(ev.sum()))
    pe = lambda d: {k: v[ev] for k, v in d.items()}
    stats("oracle32 vs float64", pe(pick(ref)), pe(ref64), out)
    for name in ("F32", "FAST"):
        stats("GPU " + name + " vs float64", pe(pick(G[name])), pe(ref64), out)

    # ---- 3. stage isolation: each stage of each level re-evaluated by the oracle on the GPU's own inputs --------------------
    out.append("\n## 3. stage isolation (oracle fp32 applied to the GPU's own stage inputs), per precision, over the %d-ray subset" % len(sub))
    b = {k: torch.from_numpy(np.ascontiguousarray(v[idx[sub]])) for k, v in full.items()}
    _, s_to_t = orc.construct_ray_warps(b["near"], b["far"], mc.power_lambda)
    enc = orc.make_encoders(sd, mc)
    sdt = orc.to_torch_sd(sd)
    names = [(f"prop_mlp_{i}", mc.prop_cfg(i)) for i in range(mc.num_levels - 1)] + [("nerf_mlp", mc.nerf_mlp)]
    for pname in ("F32", "FAST"):
        gh = [{k: torch.from_numpy(v[sub]) for k, v in lv.items()} for lv in GH[pname]]
        out.append(f"  precision {pname}:")
        prod = 1
        for li, S in enumerate(samples):
            if li == 0:
                sd_prev = torch.cat([torch.zeros_like(b["near"]), torch.ones_like(b["far"])], -1)
                w_prev = torch.ones_like(b["near"])
            else:
                dil = mc.dilation_bias + mc.dilation_multiplier / prod
                sd_prev, w_prev = orc.max_dilate_weights(gh[li - 1]["sdist"], gh[li - 1]["weights"], dil, (0., 1.), True)
                sd_prev, w_prev = sd_prev[..., 1:-1], w_prev[..., 1:-1]
            prod *= S
            logits = torch.where(sd_prev[..., 1:] > sd_prev[..., :-1], torch.log(w_prev), torch.full_like(w_prev, -torch.inf))
            s_o = orc.sample_intervals(sd_prev, logits, S, (0., 1.))
            t_o = s_to_t(gh[li]["sdist"])
            means, stds = orc.cast_rays(gh[li]["tdist"], b["origins"], b["directions"], b["radii"], b["base_x"], b["base_y"], std_scale=mc.std_scale)
            prefix, cfg = names[li]
            res = orc.mlp_forward(sdt, prefix, cfg, enc[prefix], means, stds, b["viewdirs"])
            w_o = orc.compute_alpha_weights(gh[li]["density"], gh[li]["tdist"], b["directions"], mc.opaque_background)
            ds = (s_o - gh[li]["sdist"]).abs()
            dt_u = ulps(t_o.numpy(), gh[li]["tdist"].numpy())
            dd = (res["density"] - gh[li]["density"]).abs()
            rel = dd / (res["density"].abs() + 1e-3)
            dw = (w_o - gh[li]["weights"]).abs()
            out.append(f"    level {li} (S={S}): resample |ds| mean {ds.mean():.2e} max {ds.max():.2e} | warp ulps mean {dt_u.mean():.2f} max {dt_u.max():.0f} | "
                       f"density |d| mean {dd.mean():.2e} max {dd.max():.2e} rel mean {rel.mean():.2e} rel max {rel.max():.2e} "
                       f"(density max {res['density'].max():.1f}) | alpha-weights |dw| mean {dw.mean():.2e} max {dw.max():.2e}")

    # ---- 4. trace of the worst rays ------------------------------------------------------------------------------------------
    out.append("\n## 4. trace of the worst rays (precision FAST): where the GPU's sample positions leave the oracle's")
    order = np.argsort(-D["FAST"])[:args.worst]
    for rk in order:
        line = [f"  ray {int(idx[rk])}: depth GPU {G['FAST']['depth'][rk]:.6f} oracle32 {ref['depth'][rk]:.6f} |d| {D['FAST'][rk]:.3e}"
                f"  (F32: {D['F32'][rk]:.3e}, MIXED: {D['MIXED'][rk]:.3e})"]
        if rk in sub:
            j = int(np.where(sub == rk)[0][0])
            line.append(f" float64 {ref64['depth'][j]:.6f} -> oracle32 is {abs(ref['depth'][rk] - ref64['depth'][j]):.3e} from it, GPU {abs(G['FAST']['depth'][rk] - ref64['depth'][j]):.3e}")
        out.append("".join(line))
        for li, S in enumerate(samples):
            gs, os_ = GH["FAST"][li]["sdist"][rk], refh[li]["sdist"][rk]
            u = ulps(gs, os_)
            bad = np.where(u > 4)[0]
            gd, od = GH["FAST"][li]["density"][rk], refh[li]["density"][rk]
            gw, ow = GH["FAST"][li]["weights"][rk], refh[li]["weights"][rk]
            k = int(np.argmax(np.abs(gw - ow)))
            out.append(f"      level {li}: sdist max |d| {np.abs(gs - os_).max():.2e} ({u.max():.0f} ulp), fenceposts > 4 ulp: {len(bad)}"
                       f"{'' if not len(bad) else f' (first at {int(bad[0])}: {gs[bad[0]]:.8f} vs {os_[bad[0]]:.8f})'}; "
                       f"density max |d| {np.abs(gd - od).max():.3e} (max density {od.max():.1f}); weights max |d| {np.abs(gw - ow).max():.3e} at sample {k} "
                       f"(GPU {gw[k]:.4f} oracle {ow[k]:.4f}; top weight oracle {ow.max():.4f} at {int(ow.argmax())}, GPU {gw.max():.4f} at {int(gw.argmax())})")
        # bimodality: how much of the final weight mass sits in the two largest clusters?
        ow, ot = refh[-1]["weights"][rk], refh[-1]["tdist"][rk]
        tm = 0.5 * (ot[1:] + ot[:-1])
        top = np.argsort(-ow)[:4]
        out.append("      oracle final-level weight mass: " + ", ".join(f"t={tm[k]:.4f}:w={ow[k]:.3f}" for k in top)
                   + f"; t range of samples [{ot[0]:.4f}, {ot[-1]:.4f}]")
    print("\n".join(out))


if __name__ == "__main__":
    main()
