"""Diagnostic: fused training MLP (nlr_mlp_train_*) against torch autograd through the same level's Linear stack: per-tensor errors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import config as nconfig, lidar as nlidar, training, weights as nweights
wl, S = (sys.argv[1] if len(sys.argv) > 1 else "C2"), 32
mc = nconfig.workload(wl, 12)
sd = nweights.synth_state_dict(mc, seed=5, trained_like=False)
for k in sd:
    if k.endswith("encoder.embeddings"):
        sd[k] = (sd[k] * 3e3).astype(np.float32)
b = nlidar.synthetic_sweep(width=6, seed=5, beams=nlidar.LIDAR_ANGLES[::8])
N = b["origins"].shape[0]
rng = np.random.default_rng(0)
tdist = torch.from_numpy(np.sort(rng.uniform(0.01, 1.5, (N, S + 1)).astype(np.float32), axis=-1)).cuda()
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
cfg = mc.nerf_mlp
ref = training.TrainableNerfLevel(cfg).load_reference(sd).cuda()
fus = training.TrainableNerfLevel(cfg, fused_mlp=True).load_reference(sd).cuda()
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["density", "rgb", "semantic", "intensity"]
cot = {"density": rng.normal(size=(N, S)), "rgb": rng.normal(size=(N, S, 3))}
if cfg.use_semantic: cot["semantic"] = rng.normal(size=(N, S, cfg.class_num))
if cfg.use_intensity: cot["intensity"] = rng.normal(size=(N, S))
cot = {k: torch.from_numpy(v.astype(np.float32)).cuda() for k, v in cot.items() if k in which}
store = {}
orig = training.encode_features
outs = {}
for name, lvl in (("ref", ref), ("fus", fus)):
    def keep(*a, **k):
        f = orig(*a, **k); f.retain_grad(); store[name] = f; return f
    training.encode_features = keep
    if name == "ref":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_training import _level_forward_bf16_operands
        o = _level_forward_bf16_operands(lvl, batch, tdist)
    else:
        o = lvl(batch, tdist)
    sum((o[k] * cot[k]).sum() for k in cot).backward()
    outs[name] = {k: v.detach().float().cpu().numpy() for k, v in o.items()}
training.encode_features = orig
def rep(name, got, want):
    scale = max(float(np.abs(want).max()), 1e-30)
    print(f"{name:40s} max|err|/max|ref| {np.abs(got - want).max() / scale:9.2e}   norm err {np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30):9.2e}   scale {scale:9.2e}")
for k in outs["ref"]: rep("out " + k, outs["fus"][k], outs["ref"][k])
rep("d features", store["fus"].grad.cpu().numpy(), store["ref"].grad.cpu().numpy())
for (name, p), (_, pf) in zip(ref.named_parameters(), fus.named_parameters()):
    rep("grad " + name, pf.grad.float().cpu().numpy(), p.grad.float().cpu().numpy())
