"""Oracle (test infrastructure): deterministic synthetic tensors shared by the golden generator and the tests.

numpy PCG64 streams only (stable across numpy releases), so a fixture needs to store just
(names, shapes, seed) instead of megabytes of weights.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Iterable, Tuple

import numpy as np
import torch


def synth_params(names_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> "OrderedDict[str, torch.Tensor]":
    """ndim>=2: U(+-1/sqrt(fan_in)) (so every zero_module of the reference is re-randomised —
    SURVEY.md headline fact 3); 1-D '.weight' (GroupNorm/LayerNorm gamma): 1 + 0.1 N(0,1);
    1-D '.bias': 0.1 N(0,1).  One PCG64 stream, consumed in the given order."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in names_shapes:
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape))
        if len(shape) >= 2:
            b = np.float32(1.0 / math.sqrt(int(np.prod(shape[1:]))))
            a = (rng.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * b
        elif name.endswith(".weight"):
            a = np.float32(1.0) + np.float32(0.1) * rng.standard_normal(n, dtype=np.float32)
        else:
            a = np.float32(0.1) * rng.standard_normal(n, dtype=np.float32)
        sd[name] = torch.from_numpy(a.astype(np.float32).reshape(shape))
    return sd


def randn(shape, seed: int) -> torch.Tensor:
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(tuple(shape), dtype=np.float32))


def cond_image(shape, seed: int) -> torch.Tensor:
    """cond ~ N(0,1) clipped to [-1,1]."""
    return randn(shape, seed).clamp(-1, 1)
