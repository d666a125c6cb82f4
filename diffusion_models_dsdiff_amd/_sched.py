"""Host side of dsd_sample: packs per-iteration fp32 coefficient rows and drives the device loop."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from ._lib import DSD_NCOEF, DsdSchedule, check, dptr, lib, stream_ptr


class Schedule:
    """Owns the host arrays a dsd_schedule points to.  Row k = k-th executed iteration (largest t first)."""

    def __init__(self, mode: int, pred: int, coef: np.ndarray, t_model: np.ndarray, nonzero: np.ndarray,
                 learned_range: bool = False, clip_denoised: bool = True, eta: float = 0.0):
        steps = int(coef.shape[0])
        assert coef.shape == (steps, DSD_NCOEF) and t_model.shape == (steps,) and nonzero.shape == (steps,)
        self.coef = np.ascontiguousarray(coef, dtype=np.float32)
        self.t_model = np.ascontiguousarray(t_model, dtype=np.float32)
        self.nonzero = np.ascontiguousarray(nonzero, dtype=np.int32)
        self.c = DsdSchedule()
        self.c.steps, self.c.mode, self.c.pred = steps, int(mode), int(pred)
        self.c.learned_range, self.c.clip_denoised, self.c.eta = int(learned_range), int(clip_denoised), float(eta)
        self.c.coef = self.coef.ctypes.data_as(C.POINTER(C.c_float))
        self.c.t_model = self.t_model.ctypes.data_as(C.POINTER(C.c_float))
        self.c.nonzero = self.nonzero.ctypes.data_as(C.POINTER(C.c_int32))

    @property
    def steps(self) -> int:
        return int(self.c.steps)

    def with_pred(self, pred: int) -> "Schedule":
        """The same schedule for a network output of another kind (DSD_PRED_*): the denoised_fn hook hands the update kernel
        an x_start it has already formed and post-processed."""
        return Schedule(int(self.c.mode), int(pred), self.coef, self.t_model, self.nonzero, bool(self.c.learned_range),
                        bool(self.c.clip_denoised), float(self.c.eta))


def find_unet(model):
    """Locate the native DSUnetModel behind the object the reference passes as ``model``
    (DiffusionWrapper.diffusion_model, ddpm.py:1323; or the U-Net itself)."""
    from .UNet_DS_Diff.model import DSUnetModel
    if isinstance(model, DSUnetModel):
        return model
    inner = getattr(model, "diffusion_model", None)
    if isinstance(inner, DSUnetModel):
        return inner
    inner = getattr(getattr(model, "model", None), "diffusion_model", None)
    if isinstance(inner, DSUnetModel):
        return inner
    return None


def _seed_from_torch() -> int:
    """Philox seed drawn from torch's CPU generator so torch.manual_seed() makes sampling reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


@torch.no_grad()
def run_device_loop(unet, sched: Schedule, x_T: torch.Tensor, cond: torch.Tensor,
                    step_noise: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                    first_step: int = 0, n_steps: int = 0) -> torch.Tensor:
    """x_T [B,1,H,W], cond [B,Cc,H,W] (CUDA fp32).  Returns x after the selected iterations."""
    if not x_T.is_cuda:
        raise RuntimeError("sampling runs on the MI355X only (no CPU fallback): x_T is on the CPU")
    unet.sync_params()
    x = x_T.detach().float().contiguous().clone()
    cond = cond.detach().float().contiguous()
    B, Cx, H, W = x.shape
    assert Cx == 1 and cond.shape[0] == B and cond.shape[2:] == x.shape[2:]
    if step_noise is not None:
        step_noise = step_noise.detach().float().contiguous()
        assert step_noise.shape == (sched.steps, B, 1, H, W), "step_noise must be [steps,B,1,H,W]"
    if seed is None:
        seed = _seed_from_torch()
    check(lib().dsd_sample(unet._h, C.byref(sched.c), dptr(cond), cond.shape[1], dptr(x), dptr(step_noise),
                           C.c_uint64(seed), B, H, W, first_step, n_steps, stream_ptr()))
    return x


@torch.no_grad()
def sampler_update(sched: Schedule, k: int, model_out: torch.Tensor, x: torch.Tensor,
                   noise: Optional[torch.Tensor], seed: int = 0, want_x0: bool = False):
    """One fused update (dsd_op_sampler_update); x is updated in place."""
    B, Cx, H, W = x.shape
    if Cx > 1:
        # multi-channel states (latents): the update is elementwise, so [B,C,H,W] is B*C one-channel images — except for
        # the learned-range variance, whose model output interleaves mean and variance channels per sample
        if sched.c.learned_range:
            raise NotImplementedError("learned-range variance on multi-channel states is not on the sampling hot path")
        B = B * Cx
    x0 = torch.empty_like(x) if want_x0 else None
    check(lib().dsd_op_sampler_update(C.byref(sched.c), k, dptr(model_out.float().contiguous()), dptr(x),
                                      dptr(noise.float().contiguous()) if noise is not None else None,
                                      C.c_uint64(seed), B, H, W, dptr(x0), stream_ptr()))
    return x0
