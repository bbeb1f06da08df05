"""Where does the hash-grid backward spend its time?  Levels in isolation, random vs ray-structured points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip.gridencoder import GridEncoder
dev = "cuda"
torch.manual_seed(0)
N, S = 32768, 128
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xr = torch.rand(N * S, 3, device=dev) * 2 - 1
o = torch.zeros(N, 1, 3, device=dev); d = torch.nn.functional.normalize(torch.randn(N, 1, 3, device=dev), dim=-1)
t = torch.linspace(0.02, 0.98, S, device=dev)[None, :, None]
xs = (o + t * d).reshape(-1, 3)          # rays through the origin: heavy sharing near the centre
for name, x in (("random points", xr), ("points along 32768 rays", xs)):
    for cfgname, kw in (("3 dense levels (16..64)", dict(num_levels=3, desired_resolution=64)),
                        ("10 levels (16..8192)", dict(num_levels=10, desired_resolution=8192))):
        enc = GridEncoder(input_dim=3, level_dim=4, base_resolution=16, log2_hashmap_size=21, **kw).to(dev)
        g = torch.ones(x.shape[0], enc.output_dim, device=dev)
        def fwd(): return enc(x, bound=1)
        y = fwd()
        def bwd():
            enc.embeddings.grad = None
            y2 = enc(x, bound=1); y2.backward(g)
        tf = timeit(fwd); tb = timeit(bwd) - tf
        print(f"{name:26s} {cfgname:26s}: forward {tf:7.2f} ms, backward {tb:7.2f} ms")
