"""Training-side operators on the fused path (scope row f-3): differentiable compositing and the hash-decay loss.

With `gridencoder.GridEncoder` (HIP forward + backward, Z/gridencoder/grid.py:24-89) these are the ends of the training
graph of ZI/train.py:272-281,459: hash-grid features in, composited ray outputs and the regulariser out.  The MLP between
them is `torch.nn.functional.linear` on the GPU here (the fused MFMA kernel is inference-only; its backward is the open
part of row f-3), and proposal resampling carries no gradient in the reference either (`Model.stop_level_grad`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


class _Composite(torch.autograd.Function):
    """compute_alpha_weights + volumetric_rendering (ZI/render.py:170-252): nlr_composite_level forward,
    nlr_composite_backward backward.  Per-sample inputs in the reference's layout ([N,S,3], [N,S,K], [N,S])."""

    @staticmethod
    def forward(ctx, density, tdist, dirs, rgbs, semantic, intensity, opaque_background, bg):
        if not density.is_cuda:
            raise RuntimeError("composite: density must be a CUDA tensor (no CPU fallback)")
        n, S = density.shape
        dev, f32 = density.device, torch.float32
        d = density.contiguous().float()
        td = tdist.contiguous().float()
        dr = dirs.contiguous().float()
        rgb_cm = rgbs.permute(2, 0, 1).contiguous().float() if rgbs is not None else None       # [3,N,S]
        sem_cm = semantic.permute(2, 0, 1).contiguous().float() if semantic is not None else None  # [K,N,S]
        it = intensity.reshape(n, S).contiguous().float() if intensity is not None else None
        K = 0 if semantic is None else semantic.shape[-1]
        out = _lib.NlrOut()
        res = {"rgb": torch.empty(n, 3, device=dev), "depth": torch.empty(n, device=dev), "acc": torch.empty(n, device=dev)}
        if K:
            res["semantic"] = torch.empty(n, K, device=dev)
        if it is not None:
            res["intensity"] = torch.empty(n, device=dev)
        for k, t in res.items():
            setattr(out, k, t.data_ptr())
        weights = torch.empty(n, S, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().nlr_composite_level(_lib.ptr(d), _lib.ptr(td), _lib.ptr(dr), _lib.ptr(rgb_cm), _lib.ptr(sem_cm), _lib.ptr(it), None,
                                                None, n, S, K, int(opaque_background), float(bg), 0, 0.0, _lib.ptr(weights), C.byref(out), None,
                                                _lib.current_stream())
        _lib.check(rc, "nlr_composite_level")
        ctx.save_for_backward(d, td, dr, rgb_cm, sem_cm, it)
        ctx.meta = (n, S, K, int(opaque_background), float(bg))
        sem_out = res.get("semantic", torch.zeros(n, 0, device=dev))
        int_out = res.get("intensity", torch.zeros(0, device=dev))
        return res["rgb"], res["depth"], sem_out, int_out, res["acc"], weights

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_sem, g_int, g_acc, g_w):
        d, td, dr, rgb_cm, sem_cm, it = ctx.saved_tensors
        n, S, K, opaque, bg = ctx.meta
        dev = d.device
        c = lambda t: None if t is None else t.contiguous().float()
        g_rgb, g_depth, g_acc, g_w = c(g_rgb), c(g_depth), c(g_acc), c(g_w)
        g_sem = c(g_sem) if K else None
        g_int = c(g_int) if it is not None else None
        dd = torch.empty(n, S, device=dev)
        d_rgb = torch.empty(3, n, S, device=dev) if rgb_cm is not None else None
        d_sem = torch.empty(K, n, S, device=dev) if K else None
        d_int = torch.empty(n, S, device=dev) if it is not None else None
        with torch.cuda.device(dev):
            rc = _lib.lib().nlr_composite_backward(_lib.ptr(d), _lib.ptr(td), _lib.ptr(dr), _lib.ptr(rgb_cm), _lib.ptr(sem_cm), _lib.ptr(it), n, S, K,
                                                   opaque, bg, _lib.ptr(g_rgb), _lib.ptr(g_depth), _lib.ptr(g_sem), _lib.ptr(g_int), _lib.ptr(g_acc),
                                                   _lib.ptr(g_w), _lib.ptr(dd), _lib.ptr(d_rgb), _lib.ptr(d_sem), _lib.ptr(d_int), _lib.current_stream())
        _lib.check(rc, "nlr_composite_backward")
        return (dd, None, None, None if d_rgb is None else d_rgb.permute(1, 2, 0), None if d_sem is None else d_sem.permute(1, 2, 0), d_int,
                None, None)


def volumetric_render(density, tdist, dirs, rgbs, semantic=None, intensity=None, opaque_background=True, bg=1.0) -> Dict[str, torch.Tensor]:
    """Differentiable `weights = compute_alpha_weights(...)`, `volumetric_rendering(...)` with the reference's keys
    (`rgb`, `depth`, `semantic`, `intensity`, `acc`) plus `weights` [N,S].  Gradients: density, rgbs, semantic, intensity."""
    rgb, depth, sem, inten, acc, w = _Composite.apply(density, tdist, dirs, rgbs, semantic, intensity, opaque_background, bg)
    out = {"rgb": rgb, "depth": depth, "acc": acc, "weights": w}
    if semantic is not None:
        out["semantic"] = sem
    if intensity is not None:
        out["intensity"] = inten
    return out


class _HashDecay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embeddings, offsets_host):
        if not embeddings.is_cuda:
            raise RuntimeError("hash_decay: embeddings must be a CUDA tensor (no CPU fallback)")
        e = embeddings.contiguous().float()
        off = np.ascontiguousarray(np.asarray(offsets_host, np.int32))
        L, Cc = len(off) - 1, e.shape[1]
        ss = torch.empty(L, dtype=torch.float64, device=e.device)
        with torch.cuda.device(e.device):
            rc = _lib.lib().nlr_hash_decay_forward(_lib.ptr(e), off.ctypes.data_as(C.c_void_p), L, Cc, _lib.ptr(ss), _lib.current_stream())
        _lib.check(rc, "nlr_hash_decay_forward")
        rows = torch.from_numpy(np.diff(off).astype(np.float64)).to(e.device)
        ctx.save_for_backward(e)
        ctx.off = off
        return (ss / rows.clamp_min(1)).sum().div(L * Cc).float()

    @staticmethod
    def backward(ctx, g):
        (e,) = ctx.saved_tensors
        off = ctx.off
        grad = torch.zeros_like(e)
        with torch.cuda.device(e.device):
            rc = _lib.lib().nlr_hash_decay_backward(_lib.ptr(e), off.ctypes.data_as(C.c_void_p), len(off) - 1, e.shape[1], float(g),
                                                    _lib.ptr(grad), _lib.current_stream())
        _lib.check(rc, "nlr_hash_decay_backward")
        return grad, None


def hash_decay_loss(encoders, mult: float = 1.0) -> torch.Tensor:
    """ZI/models.py:203-223 over `nerflidar_hip.gridencoder.GridEncoder` modules (static field; `Config.obj_nodecay` keeps the
    object grids out): mult * sum_enc mean_{level,channel} mean_{rows of level} embeddings^2."""
    total = None
    for enc in encoders:
        l = _HashDecay.apply(enc.embeddings, enc._offsets_host)
        total = l if total is None else total + l
    return mult * total
