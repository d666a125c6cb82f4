"""space_timesteps / SpacedDiffusion — drop-in for Disc_diff/guided_diffusion/respace.py.

The integer timestep bookkeeping stays on the host in Python ints / float64 so it is bit-exact with
the reference (python float accumulation + round-half-even, respace.py:52-57).
"""
from __future__ import annotations

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """respace.py:7-60."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired_count = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired_count:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    n_sections = len(section_counts)
    base, extra = divmod(num_timesteps, n_sections)
    picked = []
    section_start = 0
    for sec, count in enumerate(section_counts):
        size = base + (1 if sec < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac_stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            picked.append(section_start + round(pos))
            pos += frac_stride
        section_start += size
    return set(picked)


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:63-113: keeps ``use_timesteps`` of a base process; betas re-derived from the kept alphas_cumprod."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        prev = 1.0
        kept_betas = []
        for i, acp in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                kept_betas.append(1 - acp / prev)
                prev = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(kept_betas)
        super().__init__(**kwargs)

    def _model_timestep_values(self):
        """_WrappedModel.__call__ respace.py:123-128: map_tensor[ts] (int64 gather), optional *1000/T_orig in fp32."""
        t = np.asarray(self.timestep_map, dtype=np.int64).astype(np.float32)
        if self.rescale_timesteps:
            t = t * np.float32(1000.0 / self.original_num_steps)
        return t

    def _scale_timesteps(self, t):
        return t

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self._model_timestep_values(), self.rescale_timesteps)


class _WrappedModel:
    """Callers that wrap a model themselves (the role of respace.py:116-128): the network sees the ORIGINAL process's
    timestep for loop index ``ts``.  The lookup table is the one the device loop uploads (_model_timestep_values), so both
    paths hand the network identical values: int64 indices when the timesteps are not rescaled, fp32 otherwise."""

    def __init__(self, model, timestep_values, rescale_timesteps):
        self.model = model
        self.rescale_timesteps = bool(rescale_timesteps)
        self._values = th.from_numpy(np.ascontiguousarray(timestep_values, dtype=np.float32))

    def __call__(self, x, ts, **kwargs):
        values = self._values.to(ts.device)[ts.long()]
        return self.model(x, values if self.rescale_timesteps else values.long(), **kwargs)
