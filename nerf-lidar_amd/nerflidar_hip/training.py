"""Training on the fused path (scope row f-3): the operators with HIP backward kernels and the modules that chain them.

  * differentiable compositing (`nlr_composite_level` / `nlr_composite_backward`) and the hash-decay regulariser;
  * `gridencoder.GridEncoder` (HIP forward + backward + total-variation gradient, Z/gridencoder/grid.py:24-198);
  * the NerfMLP either as torch Linear modules or through the fused MFMA forward / backward (`_FusedMLP`);
  * `TrainableNerfLevel`, `TrainablePropLevel`, `TrainableModel`: the reference's `Model.forward` with autograd (proposal
    resampling carries no gradient in the reference either, `Model.stop_level_grad`), and `training_step`: the step of
    ZI/train.py:272-459 with the loss dictionary of `nerflidar_hip.losses`.
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
    _rows: Dict = {}

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
        key = (off.tobytes(), e.device)
        rows = _HashDecay._rows.get(key)                                   # (uploaded once: a host copy per step is a synchronisation)
        if rows is None:
            rows = _HashDecay._rows[key] = torch.from_numpy(np.diff(off).astype(np.float64)).to(e.device)
        ctx.save_for_backward(e)
        ctx.off = off
        return (ss / rows.clamp_min(1)).sum().div(L * Cc).float()

    @staticmethod
    def backward(ctx, g):
        (e,) = ctx.saved_tensors
        off = ctx.off
        grad = torch.zeros_like(e)
        with torch.cuda.device(e.device):
            rc = _lib.lib().nlr_hash_decay_backward(_lib.ptr(e), off.ctypes.data_as(C.c_void_p), len(off) - 1, e.shape[1], 1.0,
                                                    _lib.ptr(grad), _lib.current_stream())
        _lib.check(rc, "nlr_hash_decay_backward")
        return grad.mul_(g), None   # (the upstream scalar stays on the device: float(g) would be a host read in every step)


def hash_decay_loss(encoders, mult: float = 1.0) -> torch.Tensor:
    """ZI/models.py:203-223 over `nerflidar_hip.gridencoder.GridEncoder` modules (static field; `Config.obj_nodecay` keeps the
    object grids out): mult * sum_enc mean_{level,channel} mean_{rows of level} embeddings^2."""
    total = None
    for enc in encoders:
        l = _HashDecay.apply(enc.embeddings, enc._offsets_host)
        total = l if total is None else total + l
    return mult * total


# ---- a trainable NerfMLP level: fused cast/contract + HIP grid op (fwd/bwd) + torch Linear stack + HIP compositing (fwd/bwd) ----------
def cast_contract(batch: Dict[str, torch.Tensor], tdist: torch.Tensor, sample_n: int = 7, sample_m: int = 3, std_scale: float = 0.35,
                  rand_deg: Optional[torch.Tensor] = None):
    """Rows a-5 + a-6 (`nlr_cast_contract`): multisample means / bound [N,S,n,3] and stds / bound [N,S,n] of the intervals of
    `tdist`, as MLP.predict_density feeds them to the encoder (ZI/models.py:965-973).  Carries no gradient (tdist is detached in
    training, Model.stop_level_grad)."""
    from .models import _RAY_KEYS
    n, S = tdist.shape[0], tdist.shape[1] - 1
    dev = tdist.device
    if not tdist.is_cuda:
        raise RuntimeError("cast_contract: tdist must be a CUDA tensor (no CPU fallback)")
    rays, keep = _lib.NlrRays(), []
    for k in _RAY_KEYS:
        t = batch[k].reshape(n, -1).contiguous().float()
        keep.append(t)
        setattr(rays, k, t.data_ptr())
    td = tdist.detach().contiguous().float()
    means = torch.empty(n, S, sample_n, 3, device=dev)
    stds = torch.empty(n, S, sample_n, device=dev)
    with torch.cuda.device(dev):
        rd = None if rand_deg is None else rand_deg.reshape(n, S, sample_n).contiguous().float()   # U[0,1) draws, render.py:150
        rc = _lib.lib().nlr_cast_contract(C.byref(rays), _lib.ptr(td), n, S, sample_n, sample_m, float(std_scale), _lib.ptr(rd), _lib.ptr(means),
                                          _lib.ptr(stds), _lib.current_stream())
    _lib.check(rc, "nlr_cast_contract")
    return means, stds


def encode_features(encoder, means: torch.Tensor, stds: torch.Tensor, re_weights: bool = True) -> torch.Tensor:
    """ZI/models.py:974-979: grid features of every multisample, erf down-weighting by footprint, mean over the multisamples.
    Differentiable with respect to `encoder.embeddings` through the HIP grid operator's backward."""
    L = encoder.num_levels
    f = encoder(means.reshape(-1, 3), bound=1).reshape(means.shape[:-1] + (L, -1))          # [N,S,n,L,C]
    if re_weights:
        gs = encoder.grid_sizes.to(means.device).float()
        w = torch.erf(1 / torch.clamp(torch.sqrt(8 * stds[..., None] ** 2 * gs ** 2), min=1e-10))
        f = (f * w[..., None]).mean(dim=-3)
    else:
        f = f.mean(dim=-3) if f.dim() > 3 else f
    return f.flatten(-2, -1)


class _EncodeFeatures(torch.autograd.Function):
    """cast_rays -> contraction -> GridEncoder -> erf re-weighting -> mean over the multisamples (ZI/models.py:965-979) as one
    operator: `nlr_encode_features_forward` (the fused kernel of the inference path) and `nlr_encode_features_backward`
    (per-point feature gradients in one kernel, then the grid operator's scatter).  Differentiable with respect to the table only: sample positions carry
    no gradient in training (`Model.stop_level_grad`)."""

    @staticmethod
    def forward(ctx, table, encoder, batch, tdist, sample_n, sample_m, std_scale, rand_deg, re_weights):
        from .models import _RAY_KEYS
        n, S = tdist.shape[0], tdist.shape[1] - 1
        dev = tdist.device
        if not (table.is_cuda and tdist.is_cuda):
            raise RuntimeError("encode_features: CUDA tensors required (no CPU fallback)")
        keep = [batch[k].reshape(n, -1).contiguous().float() for k in _RAY_KEYS]
        td = tdist.detach().contiguous().float()
        rd = None if rand_deg is None else rand_deg.reshape(n, S, sample_n).contiguous().float()
        tab = table.detach().contiguous()
        ctx.args = (encoder, keep, td, rd, int(sample_n), int(sample_m), float(std_scale), int(bool(re_weights)))
        ctx.save_for_backward(tab)
        feats = torch.empty(n * S, encoder.output_dim, device=dev)
        rays, gd = _EncodeFeatures._descs(encoder, keep, tab)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nlr_encode_features_forward(C.byref(rays), _lib.ptr(td), n, S, sample_n, sample_m, float(std_scale), _lib.ptr(rd),
                                                              C.byref(gd), int(bool(re_weights)), _lib.ptr(feats), _lib.current_stream()),
                       "nlr_encode_features_forward")
        return feats.reshape(n, S, encoder.output_dim)

    @staticmethod
    def _descs(encoder, keep, tab):
        from .models import _RAY_KEYS
        rays = _lib.NlrRays()
        for k, t in zip(_RAY_KEYS, keep):
            setattr(rays, k, t.data_ptr())
        gd = _lib.NlrGridDesc()
        gd.table, gd.table_dtype = tab.data_ptr(), {torch.float32: 0, torch.float16: 1}[tab.dtype]
        gd.num_levels, gd.level_dim = encoder.num_levels, tab.shape[1]
        gd.base_resolution = int(encoder.base_resolution)
        gd.log2_per_level_scale = float(np.log2(encoder.per_level_scale))
        gd.offsets = encoder._offsets_host.data_ptr()
        gd.gridtype, gd.align_corners, gd.interp = encoder.gridtype_id, int(bool(encoder.align_corners)), encoder.interp_id
        return rays, gd

    @staticmethod
    def backward(ctx, g):
        (tab,) = ctx.saved_tensors
        encoder, keep, td, rd, sample_n, sample_m, std_scale, re_w = ctx.args
        n, S = td.shape[0], td.shape[1] - 1
        g = g.reshape(n * S, -1).contiguous().float()
        grad = torch.zeros(tab.shape, device=tab.device, dtype=torch.float32)
        pts = torch.empty(n * S * sample_n, 3, device=tab.device)                      # scratch: unit-cube positions of the multisamples
        gpt = torch.empty(n * S * sample_n, encoder.output_dim, device=tab.device)     # scratch: their feature gradients
        rays, gd = _EncodeFeatures._descs(encoder, keep, tab)
        from .gridencoder import backward_workspace
        with torch.cuda.device(tab.device):
            ws = backward_workspace(n * S * sample_n, tab.shape[1], encoder.num_levels, float(np.log2(encoder.per_level_scale)),
                                    int(encoder.base_resolution), encoder._offsets_host, encoder.gridtype_id, encoder.align_corners, tab.device)
            _lib.check(_lib.lib().nlr_encode_features_backward_ws(C.byref(rays), _lib.ptr(td), n, S, sample_n, sample_m, std_scale, _lib.ptr(rd),
                                                                  C.byref(gd), re_w, _lib.ptr(g), _lib.ptr(pts), _lib.ptr(gpt), _lib.ptr(grad),
                                                                  _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.current_stream()),
                       "nlr_encode_features_backward")
        return (grad.to(tab.dtype),) + (None,) * 8


def encode_features_fused(encoder, batch, tdist, sample_n: int = 7, sample_m: int = 3, std_scale: float = 0.35, rand_deg=None,
                          re_weights: bool = True) -> torch.Tensor:
    """[N, S, L*C] features of the intervals of `tdist`: `cast_contract` + `encode_features` in one kernel each way."""
    return _EncodeFeatures.apply(encoder.embeddings, encoder, batch, tdist, sample_n, sample_m, std_scale, rand_deg, re_weights)


# ---- fused NerfMLP for training: nlr_mlp_train_forward / nlr_mlp_train_backward (csrc/nlr_mlp_train.hip) ---------------------------
class _FusedMLP(torch.autograd.Function):
    """The Linear stack of ZI/models.py:1116-1251 (density trunk, heads, view MLP, rgb) as two MFMA-chain kernels.  The weight
    gradients are GEMMs over the tensors those kernels save: dW_l = (d pre-activation_l)^T . (input_l), M-long reductions."""

    @staticmethod
    def forward(ctx, feats, enc, level, *params):
        plan = level._plan
        M, S = feats.shape[0], level._S
        dev = feats.device
        flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        new = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
        K = level.cfg.class_num if level.cfg.use_semantic else 0
        density, rgb = new(M), new(3, M)
        sem = new(K, M) if K else None
        inten = new(M) if level.cfg.use_intensity else None
        acts = new(M, plan.act_w, dtype=torch.bfloat16)
        f = feats.detach().contiguous().float()
        e = enc.detach().contiguous().float()
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _lib.current_stream()
            _lib.check(L.nlr_train_pack(plan.handle, _lib.ptr(flat), st), "nlr_train_pack")
            _lib.check(L.nlr_mlp_train_forward(plan.handle, _lib.ptr(f), _lib.ptr(e), M, S, _lib.ptr(density), _lib.ptr(rgb), _lib.ptr(sem),
                                               _lib.ptr(inten), _lib.ptr(acts), st), "nlr_mlp_train_forward")
        ctx.level, ctx.M = level, M
        ctx.save_for_backward(f, e, density, rgb, sem if sem is not None else density.new_empty(0),
                              inten if inten is not None else density.new_empty(0), acts)
        outs = [density, rgb]
        outs.append(sem if sem is not None else density.new_empty(0))
        outs.append(inten if inten is not None else density.new_empty(0))
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_density, g_rgb, g_sem, g_inten):
        level, M = ctx.level, ctx.M
        plan, cfg = level._plan, level.cfg
        f, e, density, rgb, sem, inten, acts = ctx.saved_tensors
        dev = f.device
        K = cfg.class_num if cfg.use_semantic else 0
        gp = lambda g, ref: None if (g is None or ref.numel() == 0) else g.contiguous().float()
        gd, gr, gs, gi = gp(g_density, density), gp(g_rgb, rgb), gp(g_sem, sem), gp(g_inten, inten)
        gacts = torch.empty(M, plan.act_w + 64, device=dev, dtype=torch.bfloat16)
        d_feat = torch.empty_like(f)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nlr_mlp_train_backward(plan.handle, M, level._S, _lib.ptr(density), _lib.ptr(rgb),
                                                         _lib.ptr(sem) if sem.numel() else None, _lib.ptr(acts), _lib.ptr(gd), _lib.ptr(gr),
                                                         _lib.ptr(gs), _lib.ptr(gi), _lib.ptr(gacts), _lib.ptr(d_feat), _lib.current_stream()),
                       "nlr_mlp_train_backward")
        if getattr(level, "_keep_debug", False):  # tests look at the kernels' raw results
            level._dbg = {k: v.detach() for k, v in dict(acts=acts, gacts=gacts, d_feat=d_feat, feats=f, enc=e, density=density, rgb=rgb,
                                                        sem=sem, inten=inten).items()}
        # ---- weight gradients: plain GEMMs over the saved tensors (f32 accumulate and result)
        W, WB, D = cfg.net_width_viewdirs, cfg.bottleneck_width, cfg.net_depth_viewdirs
        HH = (64 if K else 0) + (64 if cfg.use_intensity else 0)
        c_hid, c_hbe, c_q, c_x = 0, 64, 64 + WB, 64 + WB + HH
        a = lambda c0, n: acts[:, c0:c0 + n]
        g = lambda c0, n: gacts[:, c0:c0 + n]
        E = cfg.dim_dir_enc
        enc_s = e[:, :E].to(torch.bfloat16).repeat_interleave(level._S, dim=0)  # the bf16 values the forward chain consumed

        # [M, out]^T . [M, in_0 | in_1 | ..]: bf16 operands, f32 accumulation.  The reduction runs over M (10^5..10^6) into a
        # 256 x 256 result: as ONE library GEMM that is 16 output tiles = 16 busy CUs, so M is cut into `ck` batches (split-K as a
        # batched GEMM over strided views, no copies) whose partial results are summed in f32.
        ck = 1
        while ck < 128 and M % (2 * ck) == 0 and M // (2 * ck) >= 2048:
            ck *= 2

        def wgrad(gy, *xs):
            gb = gy.reshape(ck, M // ck, gy.shape[1]).transpose(1, 2)
            return torch.cat([torch.bmm(gb, x.reshape(ck, M // ck, x.shape[1])).float().sum(0) for x in xs], 1)

        grads = []

        ones = torch.ones(ck, 1, M // ck, device=dev, dtype=torch.bfloat16)

        def lin(gy, *xs):
            grads.append(wgrad(gy, *xs))
            # bias gradient = 1^T gy, through the same split-K batched GEMM (a column reduction of a strided [M, out] view runs at
            # a tenth of the memory bandwidth as an elementwise reduce kernel)
            grads.append(torch.bmm(ones, gy.reshape(ck, M // ck, gy.shape[1])).float().sum(0)[0])

        lin(g(c_hid, 64), f.to(torch.bfloat16))
        lin(g(c_hbe, WB), a(c_hid, 64))
        r0 = 0
        if K:
            lin(g(c_q, 64), a(c_hbe, WB))
            lin(g(plan.act_w, K), a(c_q, 64))
            r0 = 64
        if cfg.use_intensity:
            lin(g(c_q + r0, 64), a(c_hbe, WB))
            lin(g(plan.act_w + K, 1), a(c_q + r0, 64))
        lin(g(c_x, W), a(c_hbe, WB), enc_s)
        if D > 1:
            lin(g(c_x + W, W), a(c_x, W), a(c_hbe, WB), enc_s)
        for l in range(2, D):
            lin(g(c_x + l * W, W), a(c_x + (l - 1) * W, W))
        lin(g(plan.act_w + 32, 3), a(c_x + (D - 1) * W, W))
        return (d_feat, None, None) + tuple(grads)


class _TrainPlan:
    def __init__(self, cfg):
        h, n = C.c_void_p(None), C.c_uint32(0)
        F = cfg.grid_num_levels * cfg.grid_level_dim
        _lib.check(_lib.lib().nlr_train_plan_create(F, cfg.net_width_viewdirs, cfg.bottleneck_width, cfg.net_depth_viewdirs, cfg.deg_view,
                                                    cfg.class_num, int(cfg.use_semantic and not cfg.no_sem_layer), int(cfg.use_intensity),
                                                    cfg.density_bias, cfg.rgb_premultiplier, cfg.rgb_bias, cfg.rgb_padding,
                                                    C.byref(h), C.byref(n)), "nlr_train_plan_create")
        self.handle, self.n_params = h, int(n.value)
        self.act_w = int(_lib.lib().nlr_train_act_width(h))

    def __del__(self):
        try:
            _lib.lib().nlr_train_plan_destroy(self.handle)
        except Exception:
            pass


class TrainableNerfLevel(torch.nn.Module):
    """The final (NerfMLP) level as a trainable module with the reference's parameter names (`encoder.embeddings`,
    `density_layer.0.weight`, `lin_second_stage_3.bias`, `sem_layer.2.weight`, ...; ZI/models.py:847-961), so that
    `load_state_dict({k[len('nerf_mlp.'):]: v ...})` takes a reference checkpoint and the trained weights go back into
    `nerflidar_hip.models.Model` for fused inference.  forward = MLP.forward (models.py:1036-1265, inference subset:
    disable_density_normals, no GLO) on the intervals of `tdist`; `render` adds the compositing."""

    def __init__(self, cfg, table_std: float = 1e-4, fused_mlp: bool = False):
        """fused_mlp: run the Linear stack through nlr_mlp_train_forward / _backward (bf16 MFMA chains, f32 accumulation) instead
        of torch Linear modules; parameters, their names and their gradients are the same objects either way."""
        super().__init__()
        from .gridencoder import GridEncoder
        nn = torch.nn
        self.cfg = cfg
        self.fused_mlp = bool(fused_mlp)
        self.fused_encode = bool(fused_mlp)  # cast + encode + re-weight + mean as one operator (encode_features_fused)
        self._plan = None
        if self.fused_mlp and cfg.use_semantic and cfg.no_sem_layer:
            raise NotImplementedError("fused training MLP: no_sem_layer=True is not wired (use fused_mlp=False)")
        if self.fused_mlp and cfg.skip_layer_dir != 0:
            # the fused kernels wire the skip concatenation into view layer 1 (nlr_mlp_train.hip); another position has the same
            # parameter count, so nothing downstream would notice (the inference path rejects it too, nlr_api.hip)
            raise NotImplementedError(f"fused training MLP: skip_layer_dir = {cfg.skip_layer_dir} is not wired, only 0 (use fused_mlp=False)")
        self.encoder = GridEncoder(input_dim=3, num_levels=cfg.grid_num_levels, level_dim=cfg.grid_level_dim,
                                   base_resolution=cfg.grid_base_resolution, desired_resolution=cfg.grid_disired_resolution,
                                   log2_hashmap_size=cfg.grid_log2_hashmap_size, gridtype="hash", align_corners=False)
        feat = cfg.grid_num_levels * cfg.grid_level_dim
        self.density_layer = nn.Sequential(nn.Linear(feat, 64), nn.ReLU(), nn.Linear(64, cfg.bottleneck_width))
        in_rgb = cfg.bottleneck_width + cfg.dim_dir_enc
        last = in_rgb
        for i in range(cfg.net_depth_viewdirs):
            lin = nn.Linear(last, cfg.net_width_viewdirs)
            nn.init.kaiming_uniform_(lin.weight)
            self.add_module(f"lin_second_stage_{i}", lin)
            last = cfg.net_width_viewdirs + (in_rgb if i == cfg.skip_layer_dir else 0)
        self.rgb_layer = nn.Linear(last, cfg.num_rgb_channels)
        if not cfg.no_sem_layer and not cfg.fixed_semantic:  # built whether or not use_semantic reads it (ZI/models.py:954-957)
            self.sem_layer = nn.Sequential(nn.Linear(cfg.bottleneck_width, 64), nn.ReLU(), nn.Linear(64, cfg.class_num))
        if cfg.use_intensity:
            self.intensity_layer = nn.Sequential(nn.Linear(cfg.bottleneck_width, 64), nn.ReLU(), nn.Linear(64, 1))

    def load_reference(self, state_dict, prefix: str = "nerf_mlp."):
        sd = {k[len(prefix):]: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v)))
              for k, v in state_dict.items() if k.startswith(prefix) and not k.endswith(("encoder.offsets", "encoder.grid_sizes", "encoder.idx"))}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        bad = [m for m in missing if not m.startswith("encoder.") or m == "encoder.embeddings"]
        if bad or unexpected:
            raise KeyError(f"state_dict mismatch: missing {bad}, unexpected {list(unexpected)}")
        return self

    def forward(self, batch: Dict[str, torch.Tensor], tdist: torch.Tensor, sample_n: int = 7, sample_m: int = 3,
                rand_deg: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        from .objects import _pos_enc
        F = torch.nn.functional
        cfg = self.cfg
        if self.fused_encode:
            feats = encode_features_fused(self.encoder, batch, tdist, sample_n, sample_m, rand_deg=rand_deg, re_weights=cfg.re_weights)
        else:
            means, stds = cast_contract(batch, tdist, sample_n, sample_m, rand_deg=rand_deg)
            feats = encode_features(self.encoder, means, stds, cfg.re_weights)
        if self.fused_mlp:
            return self._forward_fused(batch, feats)
        x = self.density_layer(feats)
        out = {"density": F.softplus(x[..., 0] + cfg.density_bias)}
        if cfg.use_semantic:
            out["semantic"] = torch.softmax(x[..., 1:1 + cfg.class_num] if cfg.no_sem_layer else self.sem_layer(x), -1)
        if cfg.use_intensity:
            out["intensity"] = self.intensity_layer(x)[..., 0]
        enc = _pos_enc(batch["viewdirs"].reshape(x.shape[0], 3).float(), cfg.deg_view)
        h = torch.cat([x, enc[:, None, :].expand(-1, x.shape[1], -1)], dim=-1)
        inputs = h
        for i in range(cfg.net_depth_viewdirs):
            h = F.relu(getattr(self, f"lin_second_stage_{i}")(h))
            if i == cfg.skip_layer_dir:
                h = torch.cat([h, inputs], dim=-1)
        rgb = torch.sigmoid(cfg.rgb_premultiplier * self.rgb_layer(h) + cfg.rgb_bias)
        out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
        return out

    def _mlp_params(self):
        """The Linear parameters in the order of the plan's flat buffer (include/nerflidar_hip.h, section 6b)."""
        cfg = self.cfg
        mods = [self.density_layer[0], self.density_layer[2]]
        if cfg.use_semantic and not cfg.no_sem_layer:
            mods += [self.sem_layer[0], self.sem_layer[2]]
        if cfg.use_intensity:
            mods += [self.intensity_layer[0], self.intensity_layer[2]]
        mods += [getattr(self, f"lin_second_stage_{i}") for i in range(cfg.net_depth_viewdirs)] + [self.rgb_layer]
        out = []
        for m in mods:
            out += [m.weight, m.bias]
        return out

    def _forward_fused(self, batch, feats):
        from .objects import _pos_enc
        cfg = self.cfg
        n, S = feats.shape[0], feats.shape[1]
        if self._plan is None:
            with torch.cuda.device(feats.device):  # nlr_train_plan_create allocates on the current device
                self._plan = _TrainPlan(cfg)
        self._S = S
        params = self._mlp_params()
        assert sum(p.numel() for p in params) == self._plan.n_params
        enc = torch.zeros(n, 32, device=feats.device)
        enc[:, :cfg.dim_dir_enc] = _pos_enc(batch["viewdirs"].reshape(n, 3).float(), cfg.deg_view)
        density, rgb, sem, inten = _FusedMLP.apply(feats.reshape(n * S, -1), enc, self, *params)
        out = {"density": density.reshape(n, S), "rgb": rgb.reshape(3, n, S).permute(1, 2, 0)}
        if cfg.use_semantic:
            out["semantic"] = sem.reshape(-1, n, S).permute(1, 2, 0)
        if cfg.use_intensity:
            out["intensity"] = inten.reshape(n, S)
        return out

    def render(self, batch, tdist, opaque_background: bool = True, bg: float = 1.0, **kw):
        o = self.forward(batch, tdist, **kw)
        r = volumetric_render(o["density"], tdist, batch["directions"].reshape(tdist.shape[0], 3), o["rgb"], o.get("semantic"),
                              o.get("intensity"), opaque_background, bg)
        return r, o


# ---- the whole model for training: Model.forward (ZI/models.py:239-576) with autograd, and the step of train.py:272-459 -------------
class _PropDensity(torch.autograd.Function):
    """raw density of a PropMLP, `density_layer(feats)[..., 0]`, through `nlr_prop_mlp_forward` / `_backward` (the hidden units are
    recomputed in the backward: only the features are saved)."""

    @staticmethod
    def forward(ctx, feats, w1, b1, w2, b2):
        f = feats.contiguous().float()
        M, F = f.shape
        raw = torch.empty(M, device=f.device)
        ps = [t.detach().contiguous().float() for t in (w1, b1, w2, b2)]
        with torch.cuda.device(f.device):
            _lib.check(_lib.lib().nlr_prop_mlp_forward(_lib.ptr(f), *[_lib.ptr(t) for t in ps], M, F, _lib.ptr(raw), _lib.current_stream()),
                       "nlr_prop_mlp_forward")
        ctx.save_for_backward(f, *ps)
        return raw

    @staticmethod
    def backward(ctx, g):
        f, w1, b1, w2, b2 = ctx.saved_tensors
        M, F = f.shape
        g = g.contiguous().float()
        d_f = torch.empty_like(f) if ctx.needs_input_grad[0] else None
        d = [torch.empty_like(t) for t in (w1, b1, w2, b2)]
        with torch.cuda.device(f.device):
            _lib.check(_lib.lib().nlr_prop_mlp_backward(_lib.ptr(f), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(g), M, F,
                                                        _lib.ptr(d_f), *[_lib.ptr(t) for t in d], _lib.current_stream()), "nlr_prop_mlp_backward")
        return (d_f, *d)


class TrainablePropLevel(torch.nn.Module):
    """A PropMLP (ZI/models.py:PropMLP: disable_rgb) with the reference's parameter names: `encoder.embeddings`,
    `density_layer.{0,2}.{weight,bias}`."""

    def __init__(self, cfg, fused: bool = True):
        """fused: the density network through `nlr_prop_mlp_forward` / `_backward` instead of two library GEMMs with 6-8 wide
        operands; same parameters, same gradients (fp32 both ways)."""
        super().__init__()
        from .gridencoder import GridEncoder
        nn = torch.nn
        self.cfg, self.fused = cfg, bool(fused)
        self.encoder = GridEncoder(input_dim=3, num_levels=cfg.grid_num_levels, level_dim=cfg.grid_level_dim,
                                   base_resolution=cfg.grid_base_resolution, desired_resolution=cfg.grid_disired_resolution,
                                   log2_hashmap_size=cfg.grid_log2_hashmap_size, gridtype="hash", align_corners=False)
        self.density_layer = nn.Sequential(nn.Linear(cfg.grid_num_levels * cfg.grid_level_dim, 64), nn.ReLU(), nn.Linear(64, 1))

    load_reference = TrainableNerfLevel.load_reference

    def forward(self, batch, tdist, sample_n: int = 7, sample_m: int = 3, rand_deg=None) -> Dict[str, torch.Tensor]:
        if self.fused:
            feats = encode_features_fused(self.encoder, batch, tdist, sample_n, sample_m, rand_deg=rand_deg, re_weights=self.cfg.re_weights)
        else:
            means, stds = cast_contract(batch, tdist, sample_n, sample_m, rand_deg=rand_deg)
            feats = encode_features(self.encoder, means, stds, self.cfg.re_weights)
        if self.fused and feats.shape[-1] <= 16:
            l0, l2 = self.density_layer[0], self.density_layer[2]
            raw = _PropDensity.apply(feats.reshape(-1, feats.shape[-1]), l0.weight, l0.bias, l2.weight, l2.bias).reshape(feats.shape[:-1])
        else:
            raw = self.density_layer(feats)[..., 0]
        return {"density": torch.nn.functional.softplus(raw + self.cfg.density_bias)}


class TrainableObjMLP(torch.nn.Module):
    """One class's object network (ZI/models.py:MLP as `ObjMLP` configures it under the shipped gin, nuscenes_single.gin:36-44) as a
    trainable module with the reference's parameter names (`encoder.embeddings`, `density_layer.{0,2}`, `lin_second_stage_{i}`,
    `rgb_layer`): hash grid on box coordinates through the HIP operator (forward + backward), [grid | shape half of the latent] -> 64 ->
    bottleneck, view MLP on [bottleneck | pos_enc(dir) | texture half] with the skip concatenation, fixed one-hot semantic of the class.
    The differentiable twin of `objects.ObjMLP.forward`."""

    def __init__(self, cfg):
        super().__init__()
        from .gridencoder import GridEncoder
        from .weights import mlp_param_shapes
        nn = torch.nn
        self.cfg = cfg
        self.encoder = GridEncoder(input_dim=3, num_levels=cfg.grid_num_levels, level_dim=cfg.grid_level_dim,
                                   base_resolution=cfg.grid_base_resolution, desired_resolution=cfg.grid_disired_resolution,
                                   log2_hashmap_size=cfg.grid_log2_hashmap_size, gridtype="hash", align_corners=False)
        for name, (o, i), kaiming in mlp_param_shapes(cfg):
            lin = nn.Linear(i, o)
            if kaiming:
                nn.init.kaiming_uniform_(lin.weight)
            mod, _, leaf = name.partition(".")
            if leaf:  # density_layer.0 / .2: an nn.Sequential with a ReLU between, like the reference's
                if not hasattr(self, mod):
                    self.add_module(mod, nn.Sequential())
                seq = getattr(self, mod)
                while len(seq) < int(leaf):
                    seq.append(nn.ReLU())
                seq.append(lin)
            else:
                self.add_module(name, lin)

    load_reference = TrainableNerfLevel.load_reference

    def forward(self, pts: torch.Tensor, viewdirs: torch.Tensor, latent: Optional[torch.Tensor]) -> Dict[str, torch.Tensor]:
        from .objects import _pos_enc
        F = torch.nn.functional
        cfg = self.cfg
        feats = self.encoder(pts.contiguous(), bound=1)
        if latent is not None:
            feats = torch.cat([feats, latent[:, : cfg.latent_size // 2] if cfg.split_latent else latent], dim=-1)
        x = self.density_layer(feats)
        out = {"density": F.softplus(x[..., 0] + cfg.density_bias)}
        if cfg.use_semantic:
            sem = torch.zeros(x.shape[0], cfg.class_num, device=x.device)
            if cfg.class_type != 255:
                sem[:, cfg.class_type] = 1.0
            out["semantic"] = sem
        h = [x, _pos_enc(viewdirs, cfg.deg_view)]
        if cfg.split_latent:
            h.append(latent[:, cfg.latent_size // 2:])
        h = torch.cat(h, dim=-1)
        inputs = h
        for i in range(cfg.net_depth_viewdirs):
            h = F.relu(getattr(self, f"lin_second_stage_{i}")(h))
            if i == cfg.skip_layer_dir:
                h = torch.cat([h, inputs], dim=-1)
        rgb = torch.sigmoid(cfg.rgb_premultiplier * self.rgb_layer(h) + cfg.rgb_bias)
        out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
        return out


class TrainableModel(torch.nn.Module):
    """`Model` (ZI/models.py:31-576, no GLO) as a trainable module: submodules `prop_mlp_<i>` and `nerf_mlp` with the reference's
    parameter names, so a reference checkpoint loads with `load_reference` and the trained `state_dict()` goes straight into
    `nerflidar_hip.models.Model` for fused inference.  With `mc.config.instance_obj` (the shipped gin, nuscenes_single.gin:13) and
    `tracks` / `class_names` also the dynamic-object branch of models.py:401-477 in training form: `obj_mlp_<class id>` per class,
    `latent_vector_dict.obj_latent_<track>` per track (train_utils.py:459-471), evaluated on the samples inside the boxes of every level
    (detached on the proposal levels, models.py:447-449) and merged before compositing; `obj_mask` rides in the ray history for the
    interlevel term (train_utils.py:153-157).  Its `state_dict()` feeds `objects.DynamicModel` / `checkpoints.dynamic_model_from_checkpoint`.

    forward = the level loop of models.py:316-557 in training form: the sample positions come from the fused resampling kernel
    (`nlr_resample_level`; they carry no gradient, Model.stop_level_grad = True), cast / contraction from `nlr_cast_contract`,
    hash-grid features and their gradient from the HIP grid operator, the NerfMLP from torch Linear modules or the fused MFMA
    forward / backward (`fused_mlp=True`), compositing and its gradient from `nlr_composite_level` / `nlr_composite_backward`."""

    def __init__(self, mc, fused_mlp: bool = False, tracks=None, class_names=None, obj_log2_hashmap: int = 21):
        super().__init__()
        self.mc = mc
        for i in range(mc.num_levels - 1):
            self.add_module(f"prop_mlp_{i}", TrainablePropLevel(mc.prop_cfg(i), fused=fused_mlp))
        import dataclasses
        ncfg = dataclasses.replace(mc.nerf_mlp, use_semantic=mc.config.use_semantic, use_intensity=mc.config.use_intensity,
                                   no_sem_layer=mc.config.no_sem_layer)
        self.nerf_mlp = TrainableNerfLevel(ncfg, fused_mlp=fused_mlp)
        self.instance_obj = bool(mc.config.instance_obj)
        if self.instance_obj:
            from .config import obj_mlp_config
            from .objects import query_class
            if tracks is None or class_names is None:
                raise ValueError("Config.instance_obj = True needs tracks [N_obj, T, 9] and one class name per track (dataset.bboxes)")
            if mc.config.use_intensity:
                raise NotImplementedError("instance_obj with use_intensity: ObjMLP has no intensity head and the reference's merge assigns "
                                          "None into the intensity tensor (ZI/models.py:469) - not a runnable configuration")
            if mc.config.latent_size <= 0:
                raise NotImplementedError("per-instance ObjMLPs (Config.latent_size = 0) are not a runnable configuration under the shipped "
                                          "gin (ObjMLP.split_latent = True indexes latent = None, ZI/models.py:1201-1203)")
            self.register_buffer("tracks", torch.as_tensor(np.asarray(tracks, np.float32)))
            self.class_ids = [query_class(c) for c in class_names]
            self._class_list = sorted(set(self.class_ids))
            self.register_buffer("_class_rank", torch.tensor([self._class_list.index(c) for c in self.class_ids]))
            for cid in self._class_list:
                self.add_module(f"obj_mlp_{cid}", TrainableObjMLP(obj_mlp_config(cid, latent_size=mc.config.latent_size, log2_hashmap=obj_log2_hashmap,
                                                                                 use_semantic=mc.config.use_semantic)))
            self.latent_vector_dict = torch.nn.ParameterDict(
                {f"obj_latent_{t}": torch.nn.Parameter(torch.nn.init.normal_(torch.empty(mc.config.latent_size))) for t in range(len(self.class_ids))})

    def levels(self):
        return [getattr(self, f"prop_mlp_{i}") for i in range(self.mc.num_levels - 1)] + [self.nerf_mlp]

    def load_reference(self, state_dict):
        for i in range(self.mc.num_levels - 1):
            getattr(self, f"prop_mlp_{i}").load_reference(state_dict, f"prop_mlp_{i}.")
        self.nerf_mlp.load_reference(state_dict, "nerf_mlp.")
        if self.instance_obj:
            for cid in self._class_list:
                getattr(self, f"obj_mlp_{cid}").load_reference(state_dict, f"obj_mlp_{cid}.")
            with torch.no_grad():
                for t in range(len(self.class_ids)):
                    v = state_dict[f"latent_vector_dict.obj_latent_{t}"]
                    self.latent_vector_dict[f"obj_latent_{t}"].copy_(v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v)))
        return self

    def reference_state_dict(self) -> Dict[str, np.ndarray]:
        """Parameters under the reference's names (what `Model(mc, sd)` / `DynamicModel` and `checkpoints.save_checkpoint` take)."""
        skip = ("encoder.offsets", "encoder.grid_sizes", "encoder.idx", "tracks", "_class_rank")
        return {k: v.detach().cpu().numpy() for k, v in self.state_dict().items() if not k.endswith(skip)}

    def latent_reg(self, latent_reg: float = 0.001) -> torch.Tensor:
        """train.py:395-399 + train_utils.latentReg (:456-457): sum over the tracks of latent_reg * ||code||.  (sic) the reference
        builds it with `torch.tensor([...])` from Python floats: a VALUE in the loss dictionary that carries no gradient - reproduced."""
        z = [p.detach().norm() for p in self.latent_vector_dict.values()]
        return latent_reg * torch.stack(z).sum()

    # -- dynamic objects (models.py:401-477) --------------------------------------------------------------------------------------
    def _object_merge(self, o: Dict[str, torch.Tensor], rgbs: torch.Tensor, batch, tdist: torch.Tensor, box: torch.Tensor, last: bool):
        """Overwrite density / rgb / semantic of the samples inside the tracks' boxes with their ObjMLP's output.  Returns
        (density, rgbs, semantic, obj_mask).  Owner of a sample = the last track whose box holds the interval midpoint
        (`nlr_box_winner`; the reference's track loop overwrites earlier tracks).  Sample positions carry no gradient."""
        n, S = tdist.shape[0], tdist.shape[1] - 1
        dev = tdist.device
        origins = batch["origins"].reshape(n, 3).contiguous().float()
        dirs = batch["directions"].reshape(n, 3).contiguous().float()
        viewdirs = batch["viewdirs"].reshape(n, 3).contiguous().float()
        winner = torch.empty(n, S, dtype=torch.int32, device=dev)
        td = tdist.detach().contiguous()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nlr_box_winner(_lib.ptr(td), _lib.ptr(origins), _lib.ptr(dirs), _lib.ptr(box), n, S, int(box.shape[1]),
                                                 _lib.ptr(winner), _lib.current_stream()), "nlr_box_winner")
        mask = winner >= 0
        density, sem = o["density"], o.get("semantic") if last else None
        sel = mask.nonzero()
        if sel.shape[0] == 0:
            return density, rgbs, sem, mask
        ri, si = sel[:, 0], sel[:, 1]
        tr = winner[ri, si].long()
        bp = box[ri, tr]                                   # cos, sin, t_w_o (3), scale (3): obj_utils.py:158-176
        t_mid = 0.5 * (td[ri, si] + td[ri, si + 1])
        pw = t_mid[:, None] * dirs[ri] + origins[ri]
        cs, sn = bp[:, 0], bp[:, 1]
        rx = cs * pw[:, 0] - sn * pw[:, 1]
        p_all = bp[:, 5:8] * (torch.stack([rx, sn * rx + cs * pw[:, 1], pw[:, 2]], dim=-1) + bp[:, 2:5])
        vd = viewdirs[ri]
        vx = cs * vd[:, 0] - sn * vd[:, 1]
        d_all = bp[:, 5:8] * torch.stack([vx, sn * vx + cs * vd[:, 1], vd[:, 2]], dim=-1)
        d_all = d_all / torch.norm(d_all, dim=-1, keepdim=True)
        table = torch.stack([self.latent_vector_dict[f"obj_latent_{t}"] for t in range(len(self.class_ids))])
        lat_all = table[tr]
        rank = self._class_rank[tr]
        density, rgbs = density.clone(), rgbs.clone()
        sem = sem.clone() if sem is not None else None
        for r_, cid in enumerate(self._class_list):
            pick = (rank == r_).nonzero()[:, 0]
            if pick.numel() == 0:
                continue
            res = getattr(self, f"obj_mlp_{cid}")(p_all[pick], d_all[pick], lat_all[pick])
            if not last:                                   # models.py:447-449: no gradient into the object networks from proposal levels
                res = {k: v.detach() for k, v in res.items()}
            density[ri[pick], si[pick]] = res["density"]
            rgbs[ri[pick], si[pick]] = res["rgb"]
            if sem is not None and "semantic" in res:
                sem[ri[pick], si[pick]] = res["semantic"]
        return density, rgbs, sem, mask

    def forward(self, batch: Dict[str, torch.Tensor], train_frac: float = 1.0, rand: Optional[torch.Generator] = None, randomized: bool = False,
                sample_n: int = 7, sample_m: int = 3, curr_track=None):
        """-> (renderings, ray_history), one entry per level, like `Model.forward`.  randomized (or a generator in `rand`): per-ray
        jitter of the sample positions (stepfun.py:216) and per-multisample rotation (render.py:150), as `model(True, ...)` draws
        them in train.py:272."""
        mc = self.mc
        L = _lib.lib()
        n = batch["origins"].shape[0]
        dev = batch["origins"].device
        near, far = batch["near"].reshape(n).contiguous().float(), batch["far"].reshape(n).contiguous().float()
        dirs = batch["directions"].reshape(n, 3).contiguous().float()
        randomized = randomized or rand is not None
        prev_s = prev_w = None
        n_prev, prod = 0, 1.0
        renderings, history = [], []
        samples = mc.level_samples()
        box = None
        if self.instance_obj:
            if "timestamp" not in batch:
                raise RuntimeError("batch['timestamp'] is missing (ZI/models.py:315)")
            tracks = (self.tracks if curr_track is None else torch.as_tensor(curr_track, device=dev, dtype=torch.float32)).contiguous()
            ts = batch["timestamp"].reshape(-1).to(dev, torch.float32).contiguous()
            box = torch.empty(n, tracks.shape[0], 8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.nlr_track_box_params(_lib.ptr(tracks), _lib.ptr(ts), n, tracks.shape[0], tracks.shape[1], _lib.ptr(box),
                                                  _lib.current_stream()), "nlr_track_box_params")
        for li, (S, level) in enumerate(zip(samples, self.levels())):
            last = li == len(samples) - 1
            use_dil = mc.dilation_bias > 0 or mc.dilation_multiplier > 0                      # models.py:322-346
            dilation = (mc.dilation_bias + mc.dilation_multiplier * 1.0 / prod) if (li > 0 and use_dil) else 0.0
            prod *= S
            anneal = (mc.anneal_slope * train_frac) / ((mc.anneal_slope - 1) * train_frac + 1) if mc.anneal_slope > 0 else 1.0
            sdist, tdist = torch.empty(n, S + 1, device=dev), torch.empty(n, S + 1, device=dev)
            jit = torch.rand(n, device=dev, generator=rand) if randomized else None
            with torch.cuda.device(dev):
                _lib.check(L.nlr_resample_level(_lib.ptr(prev_s), _lib.ptr(prev_w), n_prev, float(dilation), float(anneal), float(mc.resample_padding), S,
                                                _lib.ptr(jit), _lib.ptr(near), _lib.ptr(far), float(mc.power_lambda), n, _lib.ptr(sdist), _lib.ptr(tdist),
                                                _lib.current_stream()), "nlr_resample_level")
            rd = torch.rand(n, S, sample_n, device=dev, generator=rand) if randomized else None
            o = level(batch, tdist, sample_n, sample_m, rand_deg=rd)
            rgbs = o["rgb"] if last else torch.zeros(n, S, 3, device=dev)   # a PropMLP renders black (models.py:1119-1122)
            obj_mask = None
            if box is not None:
                dens_m, rgbs, sem_m, obj_mask = self._object_merge(o, rgbs, batch, tdist, box, last)
                o = dict(o, density=dens_m)
                if sem_m is not None:
                    o["semantic"] = sem_m
            # background colour (models.py:488-500): the range's value if it is a point, its midpoint for a deterministic render,
            # otherwise one uniform draw per ray and channel - composited here (the kernel's background is a scalar)
            lo_bg, hi_bg = mc.bg_intensity_range
            bg_rand = None
            if lo_bg == hi_bg:
                bg = float(lo_bg)
            elif not randomized:
                bg = (lo_bg + hi_bg) / 2
            else:
                bg, bg_rand = 0.0, torch.rand(n, 3, device=dev, generator=rand) * (hi_bg - lo_bg) + lo_bg
            r = volumetric_render(o["density"], tdist, dirs, rgbs, o.get("semantic") if last else None, o.get("intensity") if last else None,
                                  bool(mc.opaque_background), bg)
            if bg_rand is not None:
                r["rgb"] = r["rgb"] + (1.0 - r["acc"]).clamp_min(0.0)[:, None] * bg_rand   # render.py:226-229
            weights = r.pop("weights")
            renderings.append(r)
            history.append(dict(sdist=sdist, tdist=tdist, weights=weights, density=o["density"]))
            if obj_mask is not None:
                history[-1]["obj_mask"] = obj_mask
            prev_s, prev_w, n_prev = sdist, weights.detach().contiguous(), S
        return renderings, history


def clip_gradients(model: torch.nn.Module, grad_max_norm: float = 0.0, grad_max_val: float = 0.0) -> None:
    """train_utils.py:243-253: clip by global norm, then by value, then - unconditionally - replace NaN / +-Inf gradients by
    0 / the largest finite values (`param.grad.nan_to_num_()`), so that one degenerate ray cannot poison the Adam state."""
    if grad_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_max_norm)
    if grad_max_val > 0:
        torch.nn.utils.clip_grad_value_(model.parameters(), grad_max_val)
    for p in model.parameters():
        if p.grad is not None:
            p.grad.nan_to_num_()


def learning_rate_decay(step: int, lr_init: float = 0.01, lr_final: float = 0.001, max_steps: int = 25000, lr_delay_steps: int = 5000,
                        lr_delay_mult: float = 1e-8) -> float:
    """ZI/math.py:54-85 with the defaults of ZI/configs.py:85-88: log-linear decay from lr_init to lr_final over max_steps, eased in
    over lr_delay_steps by `mult + (1 - mult) sin(pi/2 * step / delay)`."""
    delay = lr_delay_mult + (1 - lr_delay_mult) * float(np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))) if lr_delay_steps > 0 else 1.0
    t = float(np.clip(step / max_steps, 0, 1))
    return delay * float(np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


def create_optimizer(model: torch.nn.Module, lr_init: float = 0.01, lr_final: float = 0.001, max_steps: int = 25000, lr_delay_steps: int = 5000,
                     lr_delay_mult: float = 1e-8, adam_beta1: float = 0.9, adam_beta2: float = 0.99, adam_eps: float = 1e-15):
    """train_utils.py:256-275: Adam(betas = (0.9, 0.99), eps = 1e-15) over all parameters and the learning-rate function the loop
    writes into every param_group before each step (train.py:187-190).  Returns (optimizer, lr_fn)."""
    opt = torch.optim.Adam(model.parameters(), lr=lr_init, betas=(adam_beta1, adam_beta2), eps=adam_eps)
    return opt, (lambda step: learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult))


def training_step(model: TrainableModel, optimizer: torch.optim.Optimizer, batch: Dict[str, torch.Tensor], train_frac: float = 1.0,
                  randomized: bool = True, hash_decay_mult: float = 0.1, tv_weight: float = 0.0, grad_max_norm: float = 0.0,
                  grad_max_val: float = 0.0, latent_reg: float = 0.001, as_tensors: bool = False, **loss_kw):
    """One optimiser step as train.py:272-459 takes it: forward with random jitter, the loss dictionary (`losses.total_loss` +
    hash decay), backward through the HIP backward kernels, optional total-variation gradient on the tables (grid.py:176-198),
    gradient clipping incl. the unconditional nan_to_num_ (train_utils.clip_gradients), step.  Returns the loss terms as floats, or
    with `as_tensors` as detached device scalars: reading them is the only host synchronisation of a step (without object tracks), so a
    loop that logs every n-th step keeps the next step's launches ahead of the GPU in between."""
    from . import losses as nlosses
    optimizer.zero_grad(set_to_none=True)
    renderings, history = model(batch, train_frac=train_frac, randomized=randomized)
    terms = nlosses.total_loss(renderings, history, batch, **loss_kw)
    if hash_decay_mult > 0:  # (Config.obj_nodecay, nuscenes_single.gin:24: the object grids stay out)
        terms["hash_decay"] = hash_decay_loss([lv.encoder for lv in model.levels()], hash_decay_mult)
    if getattr(model, "instance_obj", False):
        terms["latent_reg"] = model.latent_reg(latent_reg)
    loss = sum(terms.values())
    loss.backward()
    if tv_weight > 0:
        for lv in model.levels():
            lv.encoder.grad_total_variation(tv_weight)
    clip_gradients(model, grad_max_norm, grad_max_val)
    optimizer.step()
    out = {k: v.detach() for k, v in terms.items()}
    out["loss"] = loss.detach()
    return out if as_tensors else {k: float(v) for k, v in out.items()}
