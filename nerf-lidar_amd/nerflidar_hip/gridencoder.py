"""Drop-in for the reference's `gridencoder` package on MI355X.

Mirrors Z/gridencoder/grid.py: the `_backend` functions (`grid_encode_forward`,
`grid_encode_backward`; bindings.cpp:5-7), the autograd `_grid_encode` Function (grid.py:24-90) and
the `GridEncoder` module (grid.py:96-174) with the same constructor, attributes (`output_dim,
num_levels, grid_sizes, idx, embeddings, offsets, init_std`) and `forward(inputs, bound=1)`.
All arithmetic happens in libnerflidar_hip.so; a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # gridencoder.cu:15 CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")  # gridencoder.cu:16


class _Backend:
    """Same call signatures as the pybind module `_gridencoder` (gridencoder.h:12-15)."""

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                            interp, out_layout=0):
        for n, t in (("inputs", inputs), ("embeddings", embeddings), ("outputs", outputs)):
            _require_cuda(t, n)
        if inputs.dtype != torch.float32 or outputs.dtype != torch.float32:
            raise RuntimeError("inputs/outputs must be float32 tensors")
        if embeddings.dtype not in (torch.float32, torch.float16):
            raise RuntimeError("embeddings must be a floating tensor (float32 or float16)")
        off = offsets.detach().to("cpu", torch.int32).contiguous()  # host copy: see include/nerflidar_hip.h
        rc = _lib.lib().nlr_grid_encode_forward(
            _lib.ptr(inputs), _lib.ptr(embeddings), 0 if embeddings.dtype == torch.float32 else 1,
            _lib.ptr(off), _lib.ptr(outputs), B, D, C, L, float(S), int(H), _lib.ptr(dy_dx), int(gridtype),
            int(bool(align_corners)), int(interp), int(out_layout), _lib.current_stream())
        _lib.check(rc, "grid_encode_forward")

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                             gridtype, align_corners, interp, grad_layout=0):
        for n, t in (("grad", grad), ("inputs", inputs), ("grad_embeddings", grad_embeddings)):
            _require_cuda(t, n)
        off = offsets.detach().to("cpu", torch.int32).contiguous()
        rc = _lib.lib().nlr_grid_encode_backward(
            _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(off), _lib.ptr(grad_embeddings), B, D, C, L, float(S), int(H),
            _lib.ptr(dy_dx), _lib.ptr(grad_inputs), int(gridtype), int(bool(align_corners)), int(interp),
            int(grad_layout), _lib.current_stream())
        _lib.check(rc, "grid_encode_backward")


_backend = _Backend()


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        emb = embeddings.contiguous()
        # written directly in [B, L*C] (out_layout=1): no [L,B,C] -> permute -> reshape round trip (grid.py:47,57)
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=torch.float32) if calc_grad_inputs else None
        _backend.grid_encode_forward(inputs, emb, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                                     interpolation, out_layout=1)
        ctx.save_for_backward(inputs, emb, dy_dx)
        ctx.offsets = offsets
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, dy_dx = ctx.saved_tensors
        offsets = ctx.offsets
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.contiguous().float()
        grad_embeddings = torch.zeros(embeddings.shape, device=grad.device, dtype=torch.float32)
        grad_inputs = torch.zeros_like(inputs) if dy_dx is not None else None
        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx,
                                      grad_inputs, gridtype, ctx.align_corners, interpolation, grad_layout=1)
        return grad_inputs, grad_embeddings.to(embeddings.dtype), None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear', init_std=1e-4):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.init_std = init_std

        resolutions, offsets, offset = [], [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            resolution = resolution if align_corners else resolution + 1
            params_in_level = min(self.max_params, resolution ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            resolutions.append(resolution)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        # host copy for the launcher (never moves with .cuda(); avoids a D2H sync per call)
        self._offsets_host = torch.from_numpy(np.array(offsets, dtype=np.int32))
        idx = torch.empty(offset, dtype=torch.long)
        for i in range(num_levels):
            idx[offsets[i]:offsets[i + 1]] = i
        self.register_buffer('idx', idx)
        self.register_buffer('grid_sizes', torch.from_numpy(np.array(resolutions, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-self.init_std, self.init_std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self._offsets_host, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return outputs.view(prefix_shape + [self.output_dim])
