"""Shared helpers for the tests (golden fixture access, synthetic tensors, error metrics)."""
import json
import os

import numpy as np
import torch

from oracle.synth import synth_params, randn, cond_image  # noqa: F401  (oracle = checker)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def fixture_params(g, key, prefix=""):
    """Regenerate the weights a fixture was produced with: stored (names, shapes, seed) only."""
    ns = [(n, tuple(s)) for n, s in json.loads(str(g[key + "_params"]))]
    sd = synth_params(ns, int(g[key + "_seed"]))
    return {prefix + k: v for k, v in sd.items()}


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64) if not torch.is_tensor(a) else a.double().cpu()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not torch.is_tensor(b) else b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
