"""A/B of the hash-grid scatter: scattered atomics (nlr_grid_encode_backward) against the binned scatter
(nlr_grid_encode_backward_ws) on the point sets of a training step - multisample points along LiDAR rays of the trained scene's geometry
(ray-ordered: runs of equal cells) - for the proposal grid (C = 1, L = 6, 29.4 M points = 65 536 rays x 64 samples x 7) and the NerfMLP
grid (C = 4, L = 10, 14.7 M points = 65 536 rays x 32 x 7).  Also checks that the two agree.
    python scripts/grid_bwd_ab.py [rays=65536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, scene as nscene, weights as nw
dev = "cuda"
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
b = nscene.supervise(nscene.random_lidar_rays(rays, 0, 1, dev))
L_ = _lib.lib()

def points(S, spread):
    """7 multisample points per sample, samples spread around the surface like a trained proposal (half of them within +-spread scene
    units of the hit, half uniform along the ray), in unit-cube coordinates of the contracted space (inside the unit ball: identity)."""
    g = torch.Generator(device=dev).manual_seed(S)
    n = b["origins"].shape[0]
    near = b["depth"][:, None] + (torch.rand(n, S // 2, device=dev, generator=g) - 0.5) * 2 * spread
    far = torch.rand(n, S - S // 2, device=dev, generator=g) * 0.6 + 0.008
    t = torch.sort(torch.cat([near, far], 1).clamp_min(0.008), dim=1).values
    dt = torch.diff(t, dim=1, append=t[:, -1:] + 1e-3)
    tj = t[:, :, None] + dt[:, :, None] * ((torch.arange(7, device=dev) + 0.5) / 7)[None, None, :]
    p = b["origins"][:, None, None, :] + tj[..., None] * b["directions"][:, None, None, :]
    return ((p / 2 + 1) / 2).reshape(-1, 3).contiguous()

def run(name, C, Lv, desired, S, spread):
    import math
    x = points(S, spread)
    B = x.shape[0]
    offsets, sizes, pls = nw.level_table(Lv, 16, 21, desired_resolution=desired)
    off = torch.from_numpy(np.ascontiguousarray(offsets, np.int32))
    g = torch.randn(B, Lv * C, device=dev)
    n_e = int(offsets[-1])
    Sx, H = float(math.log2(pls)), 16
    need = L_.nlr_grid_backward_workspace_bytes(B, C, Lv, Sx, H, _lib.ptr(off), 0, 0)
    if not need:
        print(f"{name}: C={C}: no binned path (atomics only)")
        return
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = {}
    for tag, w in (("atomics", None), ("binned", ws)):
        gt = torch.zeros(n_e, C, device=dev)
        def call():
            _lib.check(L_.nlr_grid_encode_backward_ws(_lib.ptr(g), _lib.ptr(x), _lib.ptr(off), _lib.ptr(gt), B, 3, C, Lv, Sx, H, None, None, 0, 0, 0, 1,
                                                      _lib.ptr(w), 0 if w is None else w.numel(), None))
        call(); torch.cuda.synchronize()
        out[tag] = gt.clone()
        t0 = time.perf_counter()
        for _ in range(5): call()
        torch.cuda.synchronize()
        out[tag + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    err = (out["atomics"] - out["binned"]).abs().max().item() / max(out["atomics"].abs().max().item(), 1e-9)
    print(f"{name}: {B/1e6:.1f} M points, C={C}, L={Lv}: scattered atomics {out['atomics_ms']:.2f} ms, binned {out['binned_ms']:.2f} ms "
          f"(workspace {need/2**30:.2f} GiB), max |difference| / max |gradient| {err:.1e}")

run("proposal grid 0 (res 512)", 1, 6, 512, 64, 0.01)
run("proposal grid 1 (res 2048)", 1, 8, 2048, 64, 0.004)
L_.nlr_debug_set(2, 1)   # NLR_DBG_BINNED_C4: let the binned path take the C = 4 grid for this comparison
run("NerfMLP grid (res 8192)", 4, 10, 8192, 32, 0.002)
L_.nlr_debug_set(2, 0)


def run_xpair(name, C, Lv, desired, S, spread):
    """The C = 4 atomic scatter: one corner per instruction (rounds 2-3) against both x-corners per instruction (nlr_grid_bwd_xpair_kernel)."""
    import math
    x = points(S, spread)
    B = x.shape[0]
    offsets, sizes, pls = nw.level_table(Lv, 16, 21, desired_resolution=desired)
    off = torch.from_numpy(np.ascontiguousarray(offsets, np.int32))
    g = torch.randn(B, Lv * C, device=dev)
    out = {}
    for tag, key in (("one corner", 1), ("x-pair", 0)):
        L_.nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, key)
        gt = torch.zeros(int(offsets[-1]), C, device=dev)
        def call():
            _lib.check(L_.nlr_grid_encode_backward(_lib.ptr(g), _lib.ptr(x), _lib.ptr(off), _lib.ptr(gt), B, 3, C, Lv, float(math.log2(pls)), 16, None, None, 0, 0, 0, 1, None))
        call(); torch.cuda.synchronize()
        out[tag] = gt.clone()
        t0 = time.perf_counter()
        for _ in range(5): call()
        torch.cuda.synchronize()
        out[tag + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    L_.nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, 0)
    err = (out["one corner"] - out["x-pair"]).abs().max().item() / max(out["one corner"].abs().max().item(), 1e-9)
    print(f"{name}: {B/1e6:.1f} M points, C={C}, L={Lv}: one corner per atomic instruction {out['one corner_ms']:.2f} ms, both x-corners {out['x-pair_ms']:.2f} ms, "
          f"max |difference| / max |gradient| {err:.1e}")

run_xpair("NerfMLP grid (res 8192), 65 536 rays x 32 samples", 4, 10, 8192, 32, 0.002)


def run_xpair_c1(name, Lv, desired, S, spread):
    """The C = 1 atomic scatter (the path without a workspace): x-pair layout against the one-corner kernel."""
    import math
    x = points(S, spread)
    B = x.shape[0]
    offsets, sizes, pls = nw.level_table(Lv, 16, 21, desired_resolution=desired)
    off = torch.from_numpy(np.ascontiguousarray(offsets, np.int32))
    g = torch.randn(B, Lv, device=dev)
    out = {}
    for tag, key in (("one corner", 1), ("x-pair", 0)):
        L_.nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, key)
        gt = torch.zeros(int(offsets[-1]), 1, device=dev)
        def call():
            _lib.check(L_.nlr_grid_encode_backward(_lib.ptr(g), _lib.ptr(x), _lib.ptr(off), _lib.ptr(gt), B, 3, 1, Lv, float(math.log2(pls)), 16, None, None, 0, 0, 0, 1, None))
        call(); torch.cuda.synchronize()
        out[tag] = gt.clone()
        t0 = time.perf_counter()
        for _ in range(5): call()
        torch.cuda.synchronize()
        out[tag + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    L_.nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, 0)
    err = (out["one corner"] - out["x-pair"]).abs().max().item() / max(out["one corner"].abs().max().item(), 1e-9)
    print(f"{name}: C=1, L={Lv}, no workspace: one corner per atomic instruction {out['one corner_ms']:.2f} ms, both x-corners {out['x-pair_ms']:.2f} ms, max |difference| / max |gradient| {err:.1e}")

run_xpair_c1("proposal grid 0 (res 512)", 6, 512, 64, 0.01)
run_xpair_c1("proposal grid 1 (res 2048)", 8, 2048, 64, 0.004)
