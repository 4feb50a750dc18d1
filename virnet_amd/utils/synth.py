"""Deterministic synthetic weights and inputs.

There are no checkpoints in the build environment (the reference keeps them in
a GitHub release, ``README.md:29``), so parity and benchmarks run on weights
that both sides can regenerate bit-identically without sharing torch RNG
state: one numpy Philox stream per ``state_dict`` key, keyed by
``(seed, crc32(key))``.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch

INPUT_SEED = 20240916  # synthetic inputs are U[0,1) from this Philox seed


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def synth_param(name: str, shape: Sequence[int], seed: int = 1234) -> np.ndarray:
    """U(-b, b) fp32; b = sqrt(3/fan_in) for >=2-D tensors (unit gain), 0.05 for biases."""
    shape = tuple(int(s) for s in shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        bound = (3.0 / fan_in) ** 0.5
    else:
        bound = 0.05
    u = _rng(seed, name).random(size=shape, dtype=np.float64)
    return ((2.0 * u - 1.0) * bound).astype(np.float32)


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Regenerate a full ``state_dict`` from ``{key: shape}``."""
    return {k: torch.from_numpy(synth_param(k, s, seed)) for k, s in shapes.items()}


def synth_images(n: int, c: int, h: int, w: int, seed: int = INPUT_SEED) -> torch.Tensor:
    """U[0,1) fp32 NCHW batch (clean image range; noise is not clipped in the reference scripts)."""
    g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, (n * 1000003 + h * 1009 + w) & 0xFFFFFFFF]))
    return torch.from_numpy(g.random(size=(n, c, h, w), dtype=np.float32))
