"""Prints error statistics of the HIP path against the golden fixtures (diagnostic; run on the GPU box)."""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model

def stat(name, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    print(f"    {name:28s} max|d|={d.max():.3e} mean|d|={d.mean():.3e} max|ref|={np.abs(b).max():.3e}")

for path in sorted(glob.glob(os.path.join(ROOT, "tests/golden/fwd_*.npz"))):
    g = dict(np.load(path))
    lg = int(g["log2_hashmap"])
    mc = nconfig.workload(str(g["workload"]), None if lg < 0 else lg)
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=bool(g["trained_like"]))
    batch_np = nlidar.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    batch = {k: torch.from_numpy(v).cuda() for k, v in batch_np.items()}
    for prec in (0, 1, 2):
        model = Model(mc, sd, precision=prec)
        rend, hist = model(False, batch, 1.0, True)
        print(os.path.basename(path), "precision", prec)
        K = g["hist0_sdist"].shape[0]
        for lvl in range(mc.num_levels):
            for k in ("sdist", "tdist", "weights", "density", "rgb", "semantic", "intensity"):
                key = f"hist{lvl}_{k}"
                if key in g:
                    got = hist[lvl][k][:K].cpu().numpy()
                    stat(key, got.reshape(g[key].shape), g[key])
            stat(f"lvl{lvl}_depth", rend[lvl]["depth"].cpu().numpy(), g[f"lvl{lvl}_depth"])
        for k in [k for k in g if k.startswith("out_")]:
            stat(k, rend[-1][k[4:]].cpu().numpy(), g[k])
        if "out_semantic" in g:
            lab = rend[-1]["semantic"].cpu().numpy().argmax(-1)
            ref = g["out_semantic"].argmax(-1)
            ss = np.sort(g["out_semantic"], -1)
            print(f"    label mismatches {int((lab != ref).sum())} / {len(ref)}; min top-2 margin {np.min(ss[:, -1] - ss[:, -2]):.3e}")
