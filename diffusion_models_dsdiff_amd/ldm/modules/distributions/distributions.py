"""DiagonalGaussianDistribution — drop-in for ldm/modules/distributions/distributions.py:24-62 on device tensors.

``sample()`` is one fused HIP kernel behind the C ABI (dsd_op_gaussian_sample): clamp of the log-variance, exp(0.5·logvar)
and mean + std·eps, eps either injected (parity runs) or drawn on the device by the library's Philox4x32-10 stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ...._lib import check, dptr, lib, stream_ptr
from ...._sched import _seed_from_torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    @torch.no_grad()
    def sample(self, noise=None, seed=None):
        """:35-37.  ``noise`` ([B,E,H,W]) / ``seed`` are extensions: injected normals or the Philox seed (default: drawn
        from torch's CPU generator, so torch.manual_seed() makes it reproducible)."""
        if self.deterministic:
            return self.mean.clone()
        p = self.parameters.float().contiguous()
        if not p.is_cuda:
            raise RuntimeError("DiagonalGaussianDistribution.sample runs on the MI355X only (no CPU fallback)")
        B, E2, H, W = p.shape
        z = torch.empty((B, E2 // 2, H, W), device=p.device, dtype=torch.float32)
        if noise is not None:
            noise = noise.to(p.device).float().contiguous()
            assert noise.shape == z.shape
        if seed is None:
            seed = _seed_from_torch()
        check(lib().dsd_op_gaussian_sample(dptr(p), dptr(noise), C.c_uint64(seed), B, E2 // 2, H, W, dptr(z), stream_ptr()))
        return z

    # The two closed forms the training loss uses (:39-58); plain torch on whatever device the moments live on.
    def kl(self, other=None):
        """KL(self || other) per sample, ``other`` = the standard normal when omitted."""
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            terms = self.mean.pow(2) + self.var - 1.0 - self.logvar
        else:
            terms = (self.mean - other.mean).pow(2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar
        return 0.5 * terms.sum(dim=[1, 2, 3])

    def nll(self, sample, dims=[1, 2, 3]):
        """Negative log-likelihood of ``sample`` under this Gaussian, summed over ``dims``."""
        if self.deterministic:
            return torch.Tensor([0.])
        return 0.5 * (float(np.log(2.0 * np.pi)) + self.logvar + (sample - self.mean).pow(2) / self.var).sum(dim=dims)

    def mode(self):
        return self.mean
