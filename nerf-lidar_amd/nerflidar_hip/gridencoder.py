"""The reference's `gridencoder` package surface on MI355X (operator row a-7).

Public names and call signatures are the reference's, because callers import them:
  `_backend.grid_encode_forward / grid_encode_backward`   pybind module, Z/gridencoder/src/bindings.cpp:5-7, gridencoder.h:12-13
  `grid_encode(...)`, `_grid_encode`                      autograd op, Z/gridencoder/grid.py:24-93
  `GridEncoder(...)`                                      nn.Module, Z/gridencoder/grid.py:96-174 (constructor keywords, the
                                                          `embeddings` parameter, buffers `offsets / idx / grid_sizes`,
                                                          attributes `output_dim / num_levels / level_dim / per_level_scale /
                                                          base_resolution / init_std`, `forward(inputs, bound=1)`)
Everything behind those names is organised differently here: one immutable `GridSpec` owns the level table (built by the
same `weights.level_table` the fused render path uses) and travels through the autograd context; the operator writes
[B, L*C] directly (no [L, B, C] staging + permute) and keeps its offsets on the host, so a call never synchronises.
All arithmetic happens in libnerflidar_hip.so; a missing library or a non-GPU tensor raises (no CPU fallback).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .weights import level_table

GRIDTYPES = ("hash", "tiled")          # ids as in gridencoder.cu:66-84 (0: hash once the dense walk overflows, 1: tiled)
INTERPOLATIONS = ("linear", "smoothstep")


def _device_operand(t: torch.Tensor, name: str) -> None:
    """The checks of gridencoder.cu:15-19 (CHECK_CUDA / CHECK_CONTIGUOUS), as RuntimeError like TORCH_CHECK."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _host_offsets(offsets) -> torch.Tensor:
    """int32 level offsets on the host: the launcher derives per-level constants from them (include/nerflidar_hip.h)."""
    if isinstance(offsets, torch.Tensor):
        return offsets.detach().to("cpu", torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(offsets, np.int32))


class _Backend:
    """Positional signatures of the pybind module `_gridencoder` (gridencoder.h:12-15) + one trailing layout switch:
    0 = the reference's [L, B, C] outputs / gradients, 1 = [B, L*C] (what `GridEncoder.forward` returns anyway)."""

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                            interp, out_layout=0):
        _device_operand(inputs, "inputs")
        _device_operand(embeddings, "embeddings")
        _device_operand(outputs, "outputs")
        if inputs.dtype != torch.float32 or outputs.dtype != torch.float32:
            raise RuntimeError("inputs/outputs must be float32 tensors")
        half = {torch.float32: 0, torch.float16: 1}.get(embeddings.dtype)
        if half is None:
            raise RuntimeError("embeddings must be a floating tensor (float32 or float16)")
        off = _host_offsets(offsets)
        with torch.cuda.device(inputs.device):  # launch on the stream of the tensors' device, not of whatever device is current
            _lib.check(_lib.lib().nlr_grid_encode_forward(
                _lib.ptr(inputs), _lib.ptr(embeddings), half, _lib.ptr(off), _lib.ptr(outputs), B, D, C, L, float(S), int(H),
                _lib.ptr(dy_dx), int(gridtype), int(bool(align_corners)), int(interp), int(out_layout),
                _lib.current_stream()), "grid_encode_forward")

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                             gridtype, align_corners, interp, grad_layout=0):
        _device_operand(grad, "grad")
        _device_operand(inputs, "inputs")
        _device_operand(grad_embeddings, "grad_embeddings")
        off = _host_offsets(offsets)
        with torch.cuda.device(inputs.device):
            ws = backward_workspace(B, C, L, S, H, off, gridtype, align_corners, inputs.device)
            _lib.check(_lib.lib().nlr_grid_encode_backward_ws(
                _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(off), _lib.ptr(grad_embeddings), B, D, C, L, float(S), int(H),
                _lib.ptr(dy_dx), _lib.ptr(grad_inputs), int(gridtype), int(bool(align_corners)), int(interp),
                int(grad_layout), _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.current_stream()), "grid_encode_backward")

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        _device_operand(inputs, "inputs")
        _device_operand(embeddings, "embeddings")
        _device_operand(grad, "grad")
        if not (inputs.dtype == embeddings.dtype == grad.dtype == torch.float32):
            raise RuntimeError("grad_total_variation runs in float32 (grid.py:176 disables autocast for it)")
        off = _host_offsets(offsets)
        with torch.cuda.device(inputs.device):
            _lib.check(_lib.lib().nlr_grad_total_variation(
                _lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(grad), _lib.ptr(off), float(weight), B, D, C, L, float(S), int(H),
                int(gridtype), int(bool(align_corners)), _lib.current_stream()), "grad_total_variation")


BINNED_SCATTER = True  # large batches scatter through bins (nlr_grid_encode_backward_ws); False: scattered atomics only (A/B, tests)


def backward_workspace(B, C, L, S, H, offsets_host, gridtype, align_corners, device):
    """The binned scatter's workspace for a B-point backward, or None when the batch is small / the grid outside its envelope."""
    if not BINNED_SCATTER or B * C < (1 << 18):
        return None
    need = _lib.lib().nlr_grid_backward_workspace_bytes(int(B), int(C), int(L), float(S), int(H), _lib.ptr(offsets_host), int(gridtype),
                                                        int(bool(align_corners)))
    return torch.empty(need, dtype=torch.uint8, device=device) if need else None


_backend = _Backend()


@dataclass(frozen=True)
class GridSpec:
    """Everything the operator needs besides the tensors: the level table and the three mode switches."""
    offsets: torch.Tensor     # host int32 [L+1]
    log2_scale: float         # S of gridencoder.cu:138
    base_resolution: int      # H
    gridtype: int = 0
    align_corners: bool = False
    interpolation: int = 0

    @property
    def levels(self) -> int:
        return int(self.offsets.numel()) - 1

    def encode(self, x01: torch.Tensor, table: torch.Tensor, want_dy_dx: bool):
        """x01 [B, D] in the unit cube -> ([B, L*C] features, dy_dx or None)."""
        B, D = x01.shape
        C = table.shape[1]
        L = self.levels
        feats = x01.new_empty(B, L * C, dtype=torch.float32)
        dy_dx = x01.new_empty(B, L * D * C, dtype=torch.float32) if want_dy_dx else None
        _backend.grid_encode_forward(x01, table, self.offsets, feats, B, D, C, L, self.log2_scale, self.base_resolution,
                                     dy_dx, self.gridtype, self.align_corners, self.interpolation, 1)
        return feats, dy_dx

    def scatter(self, grad: torch.Tensor, x01: torch.Tensor, table: torch.Tensor, dy_dx):
        """Adjoint of `encode`: (d/d table as f32 [T, C], d/d x01 or None)."""
        B, D = x01.shape
        C = table.shape[1]
        g_table = torch.zeros(table.shape, device=grad.device, dtype=torch.float32)
        g_x = torch.zeros_like(x01) if dy_dx is not None else None
        _backend.grid_encode_backward(grad, x01, table, self.offsets, g_table, B, D, C, self.levels, self.log2_scale,
                                      self.base_resolution, dy_dx, g_x, self.gridtype, self.align_corners,
                                      self.interpolation, 1)
        return g_table, g_x


class _grid_encode(Function):
    """Argument order of grid.py:26-27 so that `grid_encode(...)` call sites keep working."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        spec = offsets if isinstance(offsets, GridSpec) else GridSpec(
            _host_offsets(offsets), math.log2(per_level_scale), int(base_resolution), int(gridtype), bool(align_corners),
            int(interpolation))
        x01, table = inputs.contiguous(), embeddings.contiguous()
        feats, dy_dx = spec.encode(x01, table, bool(calc_grad_inputs))
        ctx.spec = spec
        ctx.save_for_backward(x01, table, dy_dx)
        return feats

    @staticmethod
    def backward(ctx, grad):
        x01, table, dy_dx = ctx.saved_tensors
        g_table, g_x = ctx.spec.scatter(grad.contiguous().float(), x01, table, dy_dx)
        return (g_x, g_table.to(table.dtype)) + (None,) * 7


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear', init_std=1e-4):
        super().__init__()
        offsets, sizes, per_level_scale = level_table(num_levels, base_resolution, log2_hashmap_size, per_level_scale,
                                                      desired_resolution, input_dim, align_corners)
        # plain attributes callers read (ZI/models.py:867-880, train_utils.py hash decay)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.log2_hashmap_size, self.max_params = log2_hashmap_size, 2 ** log2_hashmap_size
        self.gridtype, self.gridtype_id = gridtype, GRIDTYPES.index(gridtype)
        self.interpolation, self.interp_id = interpolation, INTERPOLATIONS.index(interpolation)
        self.align_corners, self.init_std = align_corners, init_std
        self.output_dim = num_levels * level_dim
        rows = int(offsets[-1])
        self.n_params = rows * level_dim
        self.register_buffer('offsets', torch.from_numpy(offsets.copy()))
        self.register_buffer('grid_sizes', torch.from_numpy(sizes.copy()))
        # level of every table row (hash-decay regulariser, ZI/models.py:203-223)
        self.register_buffer('idx', torch.repeat_interleave(torch.arange(num_levels), torch.from_numpy(np.diff(offsets)).long()))
        self._offsets_host = torch.from_numpy(offsets.copy())  # stays on the host when the module moves
        self._spec = GridSpec(self._offsets_host, math.log2(per_level_scale), int(base_resolution), self.gridtype_id,
                              bool(align_corners), self.interp_id)
        self.embeddings = nn.Parameter(torch.empty(rows, level_dim))
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        nn.init.uniform_(self.embeddings, -self.init_std, self.init_std)

    def extra_repr(self) -> str:
        finest = self.base_resolution * self.per_level_scale ** (self.num_levels - 1)
        return (f"{self.input_dim}-D {self.gridtype} grid, {self.num_levels} levels x {self.level_dim} channels, "
                f"resolution {self.base_resolution}..{round(finest)} (x{self.per_level_scale:.4f} per level), "
                f"table {tuple(self.embeddings.shape)}, {self.interpolation} interpolation"
                f"{', align_corners' if self.align_corners else ''}")

    def forward(self, inputs, bound=1):
        lead = inputs.shape[:-1]
        x01 = ((inputs + bound) / (2 * bound)).reshape(-1, self.input_dim)  # [-bound, bound] -> unit cube (grid.py:162)
        feats = grid_encode(x01, self.embeddings, self._spec, self.per_level_scale, self.base_resolution,
                            x01.requires_grad)
        return feats.reshape(*lead, self.output_dim)

    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """grid.py:176-198: add the total-variation gradient of the cells under `inputs` (default: B uniform random points) to
        `embeddings.grad`; call it after loss.backward() and before optimizer.step()."""
        if self.embeddings.grad is None:
            raise ValueError('embeddings.grad is None: the total-variation gradient is added to an existing gradient (call it between backward and the optimiser step)')
        table = self.embeddings
        if inputs is None:
            x01 = torch.rand(B, self.input_dim, device=table.device)
        else:
            x01 = ((inputs + bound) / (2 * bound)).reshape(-1, self.input_dim).float().contiguous()
        _backend.grad_total_variation(x01, table.detach().contiguous(), table.grad, self._offsets_host, weight, x01.shape[0],
                                      self.input_dim, table.shape[1], self.num_levels, math.log2(self.per_level_scale),
                                      self.base_resolution, self.gridtype_id, self.align_corners)
