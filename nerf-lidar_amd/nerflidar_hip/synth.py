"""Deterministic synthetic weights and inputs (no network, no checkpoints ship with the reference).

Everything is derived from a counter-based splitmix64 stream so that the golden-fixture generator
(which runs in the build container next to the reference), the CPU oracle and the GPU path
regenerate bit-identical weights from a seed instead of storing 229 MiB hash tables in fixtures.
Initialisation *ranges* follow the reference:
  - nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
    (ZI/models.py:887-889, 955-961);
  - view-MLP weights: kaiming_uniform_(a=0) -> U(-sqrt(6/fan_in), sqrt(6/fan_in)) (ZI/models.py:941);
  - hash tables: U(-init_std, init_std), init_std = 1e-4 (Z/gridencoder/grid.py:151-153), or the
    "trained-like" U(-0.1, 0.1)*... set of SURVEY section 8d so densities are non-degenerate.
"""
from __future__ import annotations

import math

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """n float32 values in [0, 1) with 24 random bits each; pure function of (seed, stream, i)."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([seed * 0x10001 + 0x1234567], dtype=np.uint64))[0]
        base = splitmix64(np.array([base ^ np.uint64(stream * 0x9E3779B1 + 77)], dtype=np.uint64))[0]
        out = np.empty(n, dtype=np.float32)
        step = 1 << 22
        for s in range(0, n, step):
            e = min(n, s + step)
            i = np.arange(s, e, dtype=np.uint64)
            z = splitmix64(base + i * np.uint64(0xD1342543DE82EF95))
            out[s:e] = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return out


def uniform(seed: int, stream: int, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(seed, stream, n)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape)


def _stream_of(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def linear_init(seed: int, name: str, out_f: int, in_f: int, kaiming: bool = False):
    """Weight [out,in] and bias [out] with the reference's init ranges (see module docstring)."""
    bw = math.sqrt(6.0 / in_f) if kaiming else 1.0 / math.sqrt(in_f)
    bb = 1.0 / math.sqrt(in_f)
    w = uniform(seed, _stream_of(name + ".weight"), (out_f, in_f), -bw, bw)
    b = uniform(seed, _stream_of(name + ".bias"), (out_f,), -bb, bb)
    return w, b


def table_init(seed: int, name: str, n_entries: int, level_dim: int, std: float) -> np.ndarray:
    return uniform(seed, _stream_of(name), (n_entries, level_dim), -std, std)
