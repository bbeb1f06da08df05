"""Timing of the training-side operators (row f-3) at the bench shapes: N = 32768 rays x S = 128 samples, K = 19 classes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import training
from nerflidar_hip.gridencoder import GridEncoder
torch.manual_seed(0)
N, S, K = 32768, 128, 19
dev = "cuda"
tdist = torch.sort(torch.rand(N, S + 1, device=dev) * 2 + 0.01, dim=-1)[0]
dens = (torch.rand(N, S, device=dev) ** 4 * 40).requires_grad_(True)
dirs = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
rgbs = torch.rand(N, S, 3, device=dev).requires_grad_(True)
sem = torch.softmax(torch.randn(N, S, K, device=dev), -1).requires_grad_(True)
inten = torch.rand(N, S, device=dev).requires_grad_(True)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def comp():
    r = training.volumetric_render(dens, tdist, dirs, rgbs, sem, inten)
    (r["rgb"].sum() + r["depth"].sum() + r["semantic"].sum() + r["intensity"].sum() + r["weights"].sum()).backward()
# kernels alone (layout conversions of the autograd wrapper excluded): call the C ABI on prepared channel-major buffers
from nerflidar_hip import _lib
import ctypes as C
rgb_cm = rgbs.detach().permute(2, 0, 1).contiguous(); sem_cm = sem.detach().permute(2, 0, 1).contiguous()
g3, g1, gk, gw = torch.ones(N, 3, device=dev), torch.ones(N, device=dev), torch.ones(N, K, device=dev), torch.ones(N, S, device=dev)
dd, drgb, dsem, dint = torch.empty(N, S, device=dev), torch.empty(3, N, S, device=dev), torch.empty(K, N, S, device=dev), torch.empty(N, S, device=dev)
L = _lib.lib(); p = _lib.ptr
def bwd_only():
    _lib.check(L.nlr_composite_backward(p(dens.detach()), p(tdist), p(dirs), p(rgb_cm), p(sem_cm), p(inten.detach()), N, S, K, 1, 1.0, p(g3), p(g1), p(gk),
                                        p(g1), p(g1), p(gw), p(dd), p(drgb), p(dsem), p(dint), _lib.current_stream()))
t_b = timeit(bwd_only)
bytes_b = 4.0 * N * S * (1 + 1 + 3 + 1 + 1 + 3 + K + 1)  # density, tdist, rgb, g_w in; d_density, d_rgb, d_sem, d_int out
print(f"nlr_composite_backward: {t_b:.3f} ms, {bytes_b / t_b / 1e6:.0f} GB/s algorithmic ({bytes_b / 1e6:.0f} MB)")
print(f"volumetric_render fwd+bwd through autograd (incl. [N,S,C] <-> channel-major copies): {timeit(comp):.3f} ms")
enc = GridEncoder(input_dim=3, num_levels=10, level_dim=4, base_resolution=16, desired_resolution=8192, log2_hashmap_size=21).to(dev)
x = torch.rand(N * S, 3, device=dev) * 2 - 1
def grid():
    enc.embeddings.grad = None
    enc(x, bound=1).sum().backward()
print(f"GridEncoder fwd+bwd, {N * S / 1e6:.1f} M points, L=10 C=4, 229 MiB table: {timeit(grid, 5):.3f} ms")
def hd():
    enc.embeddings.grad = None
    training.hash_decay_loss([enc]).backward()
print(f"hash_decay_loss fwd+bwd (229 MiB table): {timeit(hd):.3f} ms")
