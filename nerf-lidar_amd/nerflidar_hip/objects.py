"""Dynamic-object branch (scope row f-1, `Config.instance_obj=True`, the shipped gin's setting) on PyTorch-ROCm.

What the reference does (ZI/models.py:306-315,401-477; ZI/obj_utils.py): per ray, interpolate every track's box pose at the
ray's timestamp (`get_pose`); per level, transform the interval midpoints into each box frame (`box_pts`), evaluate that
track's ObjMLP on the samples that fall inside the box and overwrite the static field's per-sample density / rgb /
semantic there before compositing.  SURVEY 8a routes this branch to "the unfused PyTorch path": here the static field
still runs on the fused HIP stages (`nlr_resample_level`, `nlr_mlp_level`, `nlr_composite_level`, called level by level
instead of through `nlr_render_rays`), `nlr_box_winner` finds the owning track of every sample, the object networks are small (L=7 x C=2 grid, 64-wide trunk, 32-wide view MLP)
and run as torch ops on the GPU with the hash grid through `nlr_grid_encode_forward`.  No CPU fallback anywhere.

Only latent mode (one ObjMLP per class + one latent code per track, `Config.latent_size > 0`) without symmetry or scene
fusion is covered: inference of the shipped configuration.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, synth
from .config import MLPConfig, ModelConfig, obj_mlp_config
from .gridencoder import GridEncoder
from .models import Model, _RAY_KEYS
from .weights import grid_layout, mlp_param_shapes


def query_class(class_name: str) -> int:
    """obj_utils.py:498-509: nuScenes category -> label id (255 = no fixed label)."""
    if "human" in class_name:
        return 11
    if "truck" in class_name or "trailer" in class_name or "construction" in class_name:
        return 14
    if "bus" in class_name:
        return 15
    if "car" in class_name:
        return 13
    return 255


def synthetic_tracks(batch: Dict[str, np.ndarray], n_tracks: int = 3, n_times: int = 5, seed: int = 0, size=(0.2, 0.15, 0.15),
                     depth=(0.05, 0.3)) -> np.ndarray:
    """[N_obj, T, 9] = (center3, theta_z, wlh3, timestamp, track_id) like the dataset builds them (ZI/datasets.py:1414-1456):
    boxes placed ON rays of the sweep (so that samples fall inside), drifting and turning a little over time."""
    n = batch["origins"].shape[0]
    times = np.linspace(0.0, 1.0, n_times)
    tr = np.zeros((n_tracks, n_times, 9), np.float64)
    for k in range(n_tracks):
        ray = int(synth.uniform(seed, 8100 + k, (1,), 0, n)[0]) % n
        dep = float(synth.uniform(seed, 8200 + k, (1,), depth[0], depth[1])[0])
        c0 = batch["origins"][ray].astype(np.float64) + dep * batch["directions"][ray].astype(np.float64)
        vel = synth.uniform(seed, 8300 + k, (3,), -0.01, 0.01).astype(np.float64)
        th0 = float(synth.uniform(seed, 8400 + k, (1,), -3.0, 3.0)[0])
        wlh = np.asarray(size, np.float64) * (1.0 + 0.5 * synth.uniform(seed, 8500 + k, (3,), 0, 1).astype(np.float64))
        for i, t in enumerate(times):
            tr[k, i] = np.concatenate([c0 + vel * (t - 0.5), [th0 + 0.3 * t], wlh, [t], [k]])
    return tr.astype(np.float32)


def synthetic_timestamps(n: int, seed: int = 0) -> np.ndarray:
    return synth.uniform(seed, 8600, (n, 1), 0.0, 1.0).astype(np.float32)


# ---- geometry (torch, any device) -------------------------------------------------------------------------------------------
def get_pose(time: torch.Tensor, tracks: torch.Tensor) -> torch.Tensor:
    """obj_utils.py:431-475: blend of the two recorded poses closest in time -> [N, N_obj, 9]."""
    n, n_obj, n_info = time.shape[0], tracks.shape[0], tracks.shape[-1]
    time_diff = torch.abs(time[..., None] - tracks[:, :, -2].unsqueeze(0))
    idx = torch.sort(time_diff, dim=-1)[1][..., :2]                       # [N, N_obj, 2]
    t_rec = tracks[:, :, -2].unsqueeze(0).expand(n, -1, -1)
    t1 = torch.gather(t_rec, -1, idx[..., 0:1])
    t2 = torch.gather(t_rec, -1, idx[..., 1:2])
    w1 = (torch.abs(time.unsqueeze(-1) - t2) / (torch.abs(t1 - t2) + 1e-9)).clamp(0, 1)   # [N, N_obj, 1]
    tr = tracks.unsqueeze(0).expand(n, -1, -1, -1)
    info1 = torch.gather(tr, -2, idx[..., 0:1, None].expand(-1, -1, -1, n_info)).squeeze(-2)
    info2 = torch.gather(tr, -2, idx[..., 1:2, None].expand(-1, -1, -1, n_info)).squeeze(-2)
    return w1 * info1 + (1 - w1) * info2


def _rotate_yaw_z(p: torch.Tensor, yaw: torch.Tensor) -> torch.Tensor:
    """obj_utils.py:76-113.  (sic) y' uses the already rotated x' (lines 106-107): reproduced, it changes the numbers."""
    c, s = torch.cos(yaw), torch.sin(yaw)
    px = c * p[..., 0] - s * p[..., 1]
    py = s * px + c * p[..., 1]
    return torch.stack([px, py, p[..., 2]], dim=-1)


def box_pts(pts: torch.Tensor, viewdirs: torch.Tensor, obj_pose: torch.Tensor):
    """obj_utils.py:203-234 / world2object :116-176: pts [N,S,3], viewdirs [N,3], obj_pose [N,N_obj,9] ->
    (pts_o [N,S,N_obj,3], dirs_o [N,S,N_obj,3], intersection_map [N,S,N_obj])."""
    center, theta, wlh = obj_pose[:, None, :, :3], obj_pose[:, None, :, 3], obj_pose[:, None, :, 4:7]   # broadcast over S
    scale = 1 / (wlh / 2 + 1e-9)
    t_w_o = _rotate_yaw_z(-center, theta)
    pts_o = scale * (_rotate_yaw_z(pts[:, :, None, :].expand(-1, -1, obj_pose.shape[1], -1), theta) + t_w_o)
    dirs_o = scale * _rotate_yaw_z(viewdirs[:, None, None, :].expand(-1, pts.shape[1], obj_pose.shape[1], -1), theta)
    dirs_o = dirs_o / torch.norm(dirs_o, dim=-1, keepdim=True)
    imap = (pts_o[..., 0].abs() < 1) & (pts_o[..., 1].abs() < 1) & (pts_o[..., 2].abs() < 1)
    return pts_o, dirs_o, imap


def _pos_enc(x: torch.Tensor, deg: int) -> torch.Tensor:
    """coord.pos_enc(min_deg=0, max_deg=deg, append_identity=True) (ZI/coord.py:199-210)."""
    scales = 2.0 ** torch.arange(0, deg, device=x.device, dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(x.shape[:-1] + (-1,))
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * np.pi], dim=-1))], dim=-1)


class ObjMLP:
    """One class's object network: our `GridEncoder` (HIP op) + the Linear stack of ZI/models.py:MLP as ObjMLP configures it."""

    def __init__(self, prefix: str, cfg: MLPConfig, sd: Dict[str, np.ndarray], device):
        self.cfg, self.prefix = cfg, prefix
        dev = torch.device(device)
        self.encoder = GridEncoder(input_dim=3, num_levels=cfg.grid_num_levels, level_dim=cfg.grid_level_dim,
                                   base_resolution=cfg.grid_base_resolution, desired_resolution=cfg.grid_disired_resolution,
                                   log2_hashmap_size=cfg.grid_log2_hashmap_size, gridtype="hash", align_corners=False).to(dev)
        emb = torch.from_numpy(np.ascontiguousarray(sd[f"{prefix}.encoder.embeddings"], np.float32))
        if tuple(emb.shape) != tuple(self.encoder.embeddings.shape):
            raise ValueError(f"{prefix}.encoder.embeddings: shape {tuple(emb.shape)}, expected {tuple(self.encoder.embeddings.shape)}")
        with torch.no_grad():
            self.encoder.embeddings.copy_(emb)
        self.lin = {}
        for name, shape, _ in mlp_param_shapes(cfg):
            w, b = sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"]
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"{prefix}.{name}.weight: shape {tuple(w.shape)}, expected {tuple(shape)}")
            self.lin[name] = (torch.from_numpy(np.ascontiguousarray(w, np.float32)).to(dev),
                              torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(dev))

    def _l(self, name, x):
        w, b = self.lin[name]
        return F.linear(x, w, b)

    @torch.no_grad()
    def forward(self, pts: torch.Tensor, viewdirs: torch.Tensor, latent: Optional[torch.Tensor]):
        """ZI/models.py:1036-1265 for ObjMLP: pts [n,3] in box coordinates, unit viewdirs [n,3], latent [latent_size] (one track)
        or [n, latent_size] (a code per point: points of several tracks of this class in one call)."""
        cfg = self.cfg
        feats = self.encoder(pts.contiguous(), bound=1)
        if latent is not None:
            if latent.dim() == 1:  # one track: the reference repeats its code for every point (models.py:438)
                latent = latent[None, :].expand(feats.shape[0], -1)
            feats = torch.cat([feats, latent[:, : cfg.latent_size // 2] if cfg.split_latent else latent], dim=-1)
        x = self._l("density_layer.2", F.relu(self._l("density_layer.0", feats)))
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
            h = F.relu(self._l(f"lin_second_stage_{i}", h))
            if i == cfg.skip_layer_dir:
                h = torch.cat([h, inputs], dim=-1)
        rgb = torch.sigmoid(cfg.rgb_premultiplier * self._l("rgb_layer", h) + cfg.rgb_bias)
        out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
        return out


class DynamicModel(Model):
    """`Model` with `Config.instance_obj=True`: same constructor plus the tracks, and `forward` / `render_rays` that take a
    batch with `timestamp` [N,1].

    tracks: [N_obj, T, 9] (center3, theta_z, wlh3, timestamp, track_id), the array `Model.init_tracks` stacks from
    `dataset.bboxes[0]` (ZI/models.py:180-186); class_names: `dataset.bboxes[1]` as a list indexed by track.
    state_dict additionally carries `obj_mlp_<class id>.*` and `latent_vector_dict.obj_latent_<track>`."""

    def __init__(self, mc: ModelConfig, state_dict, tracks, class_names: Sequence[str], device="cuda:0", obj_log2_hashmap: int = 21, **kw):
        import dataclasses
        static_mc = dataclasses.replace(mc, config=dataclasses.replace(mc.config, instance_obj=False))
        super().__init__(static_mc, state_dict, device=device, **kw)
        if mc.config.use_intensity:
            raise NotImplementedError("instance_obj with use_intensity: ObjMLP has no intensity head and the reference's merge "
                                      "assigns None into the intensity tensor (ZI/models.py:469) - not a runnable configuration")
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}
        self.tracks = torch.as_tensor(np.asarray(tracks, np.float32), device=self.device)
        self.class_ids = [query_class(c) for c in class_names]
        if len(self.class_ids) != self.tracks.shape[0]:
            raise ValueError("one class name per track is required")
        lat = mc.config.latent_size
        self.obj_mlps: Dict[int, ObjMLP] = {}
        for cid in sorted(set(self.class_ids)):
            cfg = obj_mlp_config(cid, latent_size=lat, log2_hashmap=obj_log2_hashmap, use_semantic=mc.config.use_semantic)
            self.obj_mlps[cid] = ObjMLP(f"obj_mlp_{cid}", cfg, sd, self.device)
        self._class_list = sorted(self.obj_mlps)
        self._class_rank = torch.tensor([self._class_list.index(c) for c in self.class_ids], device=self.device)
        self._track_rank = torch.arange(1, len(self.class_ids) + 1, device=self.device)
        self.latents = [torch.from_numpy(np.ascontiguousarray(sd[f"latent_vector_dict.obj_latent_{t}"], np.float32)).to(self.device)
                        for t in range(len(self.class_ids))] if lat > 0 else [None] * len(self.class_ids)
        self._latent_table = torch.stack(self.latents) if lat > 0 else None
        self._objects = C.c_void_p(None)
        self._build_native(sd, lat)

    def _build_native(self, sd, lat: int) -> None:
        """Pack the object networks for the device-side branch (`nlr_objects_create`, header section 7b)."""
        n_cls = len(self._class_list)
        descs = (_lib.NlrObjClassDesc * n_cls)()
        for i, cid in enumerate(self._class_list):
            net = self.obj_mlps[cid]
            # grid + Linear stack through Model's descriptor assembly; the table is the ObjMLP encoder's own parameter
            md = self._mlp_desc(net.prefix, net.cfg, sd, torch.float32)
            md.grid.table = net.encoder.embeddings.data_ptr()
            self.tables.pop(net.prefix, None)
            descs[i].mlp = md
            descs[i].latent_size, descs[i].split_latent = lat, int(net.cfg.split_latent)
            descs[i].class_type = int(net.cfg.class_type)
        tc = np.ascontiguousarray([self._class_list.index(c) for c in self.class_ids], np.int32)
        lt = np.ascontiguousarray(self._latent_table.cpu().numpy(), np.float32) if lat > 0 else None
        od = _lib.NlrObjectsDesc()
        od.n_classes, od.classes = n_cls, descs
        od.n_tracks, od.track_class = len(self.class_ids), tc.ctypes.data
        od.latents = lt.ctypes.data if lt is not None else None
        with torch.cuda.device(self.device):
            rc = _lib.lib().nlr_objects_create(C.byref(od), C.byref(self._objects), _lib.current_stream())
        self._keep.clear()
        _lib.check(rc, "nlr_objects_create")

    def __del__(self):
        try:
            if self._objects:
                _lib.lib().nlr_objects_destroy(self._objects)
                self._objects = C.c_void_p(None)
            Model.__del__(self)
        except Exception:  # (interpreter shutdown: module globals may already be gone)
            pass

    def _workspace(self, n: int) -> torch.Tensor:
        L = _lib.lib()
        need = L.nlr_workspace_bytes(self._handle, n) + L.nlr_objects_workspace_bytes(self._objects, n, max(self.mc.level_samples()))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _render_call(self, rays, n: int, cfg, out) -> None:
        box, winners = self._dyn_call
        ws = self._workspace(n)
        wp = (_lib.c_fp * len(winners))(*[w.data_ptr() for w in winners]) if winners else None
        rc = _lib.lib().nlr_render_rays_dynamic(self._handle, self._objects, C.byref(rays), _lib.ptr(box), int(box.shape[1]), n, C.byref(cfg),
                                                C.byref(out), wp, _lib.ptr(ws), ws.numel(), _lib.current_stream())
        _lib.check(rc, "nlr_render_rays_dynamic")

    def box_params(self, timestamp: torch.Tensor, curr_track=None) -> torch.Tensor:
        """[N, n_obj, 8] world -> box constants of every ray and track at the ray's timestamp (`nlr_track_box_params`:
        get_pose + the constants of world2object, obj_utils.py:431-475,116-176)."""
        if curr_track is None:
            curr_track = getattr(self, "_track_override", None)
        tracks = self.tracks if curr_track is None else torch.as_tensor(curr_track, device=self.device, dtype=torch.float32)
        tracks = tracks.contiguous()
        ts = timestamp.reshape(-1).to(self.device, torch.float32).contiguous()
        box = torch.empty(ts.shape[0], tracks.shape[0], 8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().nlr_track_box_params(_lib.ptr(tracks), _lib.ptr(ts), ts.shape[0], tracks.shape[0], tracks.shape[1],
                                                       _lib.ptr(box), _lib.current_stream()), "nlr_track_box_params")
        return box

    @torch.no_grad()
    def render_rays(self, batch, train_frac: float = 1.0, compute_extras: bool = True, sample_n: int = 7, sample_m: int = 3,
                    want_history: bool = False, scale_factor: float = 0.0, rand_jitter=None, rand_deg=None, packed=None, curr_track=None):
        """`Model.render_rays` with the object merge of ZI/models.py:401-477 inside every level, all on the device
        (`nlr_render_rays_dynamic`): pose blend, owner map, per-class compaction and the object networks run as kernels on
        the render stream; nothing is read back.  History entries carry `obj_mask` (winner >= 0) as the reference's
        ray_results do."""
        if "timestamp" not in batch:
            raise RuntimeError("batch['timestamp'] is missing (ZI/models.py:315)")
        n = batch["origins"].shape[0]
        box = self.box_params(batch["timestamp"], curr_track)
        # -1 = "no track owns this sample": nlr_objects_apply returns before writing when there are no tracks (or no samples)
        winners = [torch.full((n, S), -1, dtype=torch.int32, device=self.device) for S in self.mc.level_samples()]
        self._dyn_call = (box, winners)
        try:
            r, hist = Model.render_rays(self, batch, train_frac, compute_extras, sample_n, sample_m, want_history, scale_factor, rand_jitter,
                                        rand_deg, packed)
        finally:
            self._dyn_call = None
        for h, w in zip(hist, winners):
            h["obj_mask"] = w >= 0
        return r, hist

    def forward(self, rand, batch, train_frac, compute_extras, zero_glo=True, sample_n=7, sample_m=3, step=0, max_step=25000,
                curr_track=None):
        """ZI/models.py:239-251: `curr_track` replaces the stored tracks for this call (models.py:307-313)."""
        self._track_override = curr_track
        try:
            return Model.forward(self, rand, batch, train_frac, compute_extras, zero_glo, sample_n, sample_m, step, max_step)
        finally:
            self._track_override = None

    __call__ = forward

    # -- the reference's level loop, stage by stage, with the object networks as torch ops: the round-1 form of this branch,
    # kept as an independent second implementation (tests compare the two; it also serves object configurations outside
    # nlr_objects_create's envelope) ---------------
    @torch.no_grad()
    def render_rays_torch(self, batch, train_frac: float = 1.0, compute_extras: bool = True, sample_n: int = 7, sample_m: int = 3,
                    want_history: bool = False, scale_factor: float = 0.0, rand_jitter=None, rand_deg=None, curr_track=None):
        if rand_jitter is not None or rand_deg is not None:
            raise NotImplementedError("DynamicModel renders deterministically (rand=False), as render_lidar does")
        if "timestamp" not in batch:
            raise RuntimeError("batch['timestamp'] is missing (ZI/models.py:315)")
        L = _lib.lib()
        mc, dev, f32 = self.mc, self.device, torch.float32
        n = batch["origins"].shape[0]
        rays, keep = _lib.NlrRays(), []
        for k in _RAY_KEYS:
            t = batch[k]
            if not t.is_cuda:
                raise RuntimeError(f"batch['{k}'] must be a CUDA tensor (no CPU fallback)")
            t = t.reshape(n, -1).contiguous().float()
            keep.append(t)
            setattr(rays, k, t.data_ptr())
        origins, dirs, viewdirs = keep[_RAY_KEYS.index("origins")], keep[_RAY_KEYS.index("directions")], keep[_RAY_KEYS.index("viewdirs")]
        near, far = keep[_RAY_KEYS.index("near")], keep[_RAY_KEYS.index("far")]
        if curr_track is None:
            curr_track = getattr(self, "_track_override", None)
        tracks = self.tracks if curr_track is None else torch.as_tensor(curr_track, device=dev, dtype=f32)
        obj_pose = get_pose(batch["timestamp"].reshape(n, 1).float().to(dev), tracks)
        # per (ray, track) constants of the world -> box map, with the reference's expressions (obj_utils.py:5-28,158-170)
        theta = obj_pose[:, :, 3]
        scale = 1 / (obj_pose[:, :, 4:7] / 2 + 1e-9)
        t_w_o = _rotate_yaw_z(-obj_pose[:, :, :3], theta)
        box_params = torch.cat([torch.cos(theta)[..., None], torch.sin(theta)[..., None], t_w_o, scale], dim=-1).contiguous()
        n_obj = obj_pose.shape[1]
        K = mc.nerf_mlp.class_num if self.config.use_semantic else 0
        new = lambda *shape, dtype=f32: torch.empty(*shape, device=dev, dtype=dtype)
        ws = torch.empty(max(int(L.nlr_workspace_bytes(self._handle, n)), 1 << 20), dtype=torch.uint8, device=dev)
        st = _lib.current_stream()
        samples = mc.level_samples()
        prev_s = prev_w = None
        n_prev, prod = 0, 1.0
        hist: List[Dict[str, torch.Tensor]] = []
        r: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(dev):
            for li, S in enumerate(samples):
                last = li == len(samples) - 1
                use_dil = mc.dilation_bias > 0 or mc.dilation_multiplier > 0                      # models.py:322-346
                dilation = (mc.dilation_bias + mc.dilation_multiplier * 1.0 / prod) if (li > 0 and use_dil) else 0.0
                prod *= S
                anneal = (mc.anneal_slope * train_frac) / ((mc.anneal_slope - 1) * train_frac + 1) if mc.anneal_slope > 0 else 1.0
                sdist, tdist = new(n, S + 1), new(n, S + 1)
                _lib.check(L.nlr_resample_level(_lib.ptr(prev_s), _lib.ptr(prev_w), n_prev, float(dilation), float(anneal),
                                                float(mc.resample_padding), S, None, _lib.ptr(near), _lib.ptr(far), float(mc.power_lambda),
                                                n, _lib.ptr(sdist), _lib.ptr(tdist), st), "nlr_resample_level")
                density = new(n, S)
                rgb = new(3, n, S) if last else None
                sem = new(K, n, S) if (last and K) else None
                _lib.check(L.nlr_mlp_level(self._handle, li, C.byref(rays), _lib.ptr(tdist), n, sample_n, sample_m, None, None,
                                           _lib.ptr(density), _lib.ptr(rgb), _lib.ptr(sem), None, _lib.ptr(ws), ws.numel(), st), "nlr_mlp_level")
                # ---- dynamic objects: overwrite the samples inside each track's box (models.py:401-477), on the rays that
                # can reach a box at all (a few % of a sweep); one host sync per level for the per-track counts
                # owner of every sample (nlr_box_winner: the last track whose box holds the interval midpoint; the reference's
                # track loop overwrites earlier tracks, models.py:415,475), then only the owned samples - 0.1-2 % of a sweep -
                # are gathered, all tracks of one class through one ObjMLP call with a latent code per point
                winner = new(n, S, dtype=torch.int32)
                _lib.check(L.nlr_box_winner(_lib.ptr(tdist), _lib.ptr(origins), _lib.ptr(dirs), _lib.ptr(box_params), n, S, n_obj,
                                            _lib.ptr(winner), st), "nlr_box_winner")
                obj_mask = winner >= 0
                sel = obj_mask.nonzero()                                                   # host sync 1: [P, 2]
                if sel.shape[0]:
                    ri, si = sel[:, 0], sel[:, 1]
                    tr = winner[ri, si].long()
                    order = torch.sort(self._class_rank[tr], stable=True)[1]
                    ri, si, tr = ri[order], si[order], tr[order]
                    per_class = torch.bincount(self._class_rank[tr], minlength=len(self._class_list)).tolist()   # host sync 2
                    # box coordinates of the owned samples, by the expressions of obj_utils.world2object (:158-176)
                    bp = box_params[ri, tr]
                    t_mid = 0.5 * (tdist[ri, si] + tdist[ri, si + 1])
                    pw = t_mid[:, None] * dirs[ri] + origins[ri]
                    cs, sn = bp[:, 0], bp[:, 1]
                    rx = cs * pw[:, 0] - sn * pw[:, 1]
                    p_all = bp[:, 5:8] * (torch.stack([rx, sn * rx + cs * pw[:, 1], pw[:, 2]], dim=-1) + bp[:, 2:5])
                    vd = viewdirs[ri]
                    vx = cs * vd[:, 0] - sn * vd[:, 1]
                    d_all = bp[:, 5:8] * torch.stack([vx, sn * vx + cs * vd[:, 1], vd[:, 2]], dim=-1)
                    d_all = d_all / torch.norm(d_all, dim=-1, keepdim=True)
                    lat_all = self._latent_table[tr] if self._latent_table is not None else None
                    idx = ri * S + si
                    lo = 0
                    for rank, cnt in enumerate(per_class):
                        if cnt == 0:
                            continue
                        sl = slice(lo, lo + cnt)
                        lo += cnt
                        o = self.obj_mlps[self._class_list[rank]].forward(p_all[sl], d_all[sl], None if lat_all is None else lat_all[sl])
                        density.view(-1)[idx[sl]] = o["density"]
                        if last:
                            rgb.view(3, -1)[:, idx[sl]] = o["rgb"].t()
                            if sem is not None:
                                sem.view(K, -1)[:, idx[sl]] = o["semantic"].t()
                weights, depth_l = new(n, S), new(n)
                out = _lib.NlrOut()
                if last:
                    r = {"rgb": new(n, 3), "depth": new(n)}
                    if K:
                        r["semantic"] = new(n, K)
                    if compute_extras:
                        for k in ("acc", "distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
                            r[k] = new(n)
                    if scale_factor > 0:
                        r["points"] = new(n, 3)
                        if K:
                            r["labels"] = new(n, dtype=torch.int32)
                    for k, t in r.items():
                        setattr(out, k, t.data_ptr())
                bg = mc.bg_intensity_range[0] if mc.bg_intensity_range[0] == mc.bg_intensity_range[1] else sum(mc.bg_intensity_range) / 2
                _lib.check(L.nlr_composite_level(_lib.ptr(density), _lib.ptr(tdist), _lib.ptr(dirs), _lib.ptr(rgb), _lib.ptr(sem), None,
                                                 _lib.ptr(far), _lib.ptr(origins), n, S, K, int(mc.opaque_background), float(bg),
                                                 int(compute_extras and last), float(scale_factor if last else 0.0), _lib.ptr(weights),
                                                 C.byref(out) if last else None, _lib.ptr(depth_l), st), "nlr_composite_level")
                h = {"depth": depth_l, "obj_mask": obj_mask}
                if want_history:
                    h.update(sdist=sdist, tdist=tdist, weights=weights, density=density)
                    if last:
                        h["rgb"] = rgb.permute(1, 2, 0)
                        if sem is not None:
                            h["semantic"] = sem.permute(1, 2, 0)
                hist.append(h)
                prev_s, prev_w, n_prev = sdist, weights, S
        return r, hist
