"""Deterministic UNet parameters shared by make_golden.py and tests/test_raydrop.py."""
import numpy as np
import torch

from nerflidar_hip import synth


def unet_fill(model, seed):
    """Deterministic parameters/buffers for a UNet (shared by the fixture generator and the tests)."""
    with torch.no_grad():
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if t.dtype.is_floating_point:
                fan = max(1, int(np.prod(t.shape[1:]))) if t.dim() > 1 else 1
                lo, hi = (0.5, 1.5) if name.endswith("running_var") else (-1.0, 1.0)
                sc = 1.0 if t.dim() <= 1 else (3.0 / fan) ** 0.5
                v = synth.uniform(seed, synth._stream_of("unet." + name), tuple(t.shape), lo, hi) * np.float32(sc)
                if name.endswith(("bn.weight",)) or (".1.weight" in name or ".4.weight" in name) and t.dim() == 1:
                    v = np.abs(v) + np.float32(0.5)
                t.copy_(torch.from_numpy(v))


