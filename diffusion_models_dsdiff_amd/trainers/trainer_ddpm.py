"""DDPMModel — the sampling surface of trainers/trainer_ddpm.py (Lightning/data/NIfTI parts are out of scope).

``sample`` / ``p_sample_loop`` (:442-459), ``p_sample`` (:461-467), ``p_mean_variance`` (:469-482) and
``apply_model`` (:484-499), executed by dsd_sample (mode B_DDPM)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._sched import Schedule, find_unet, run_device_loop
from ..ldm.models.diffusion.ddpm import DDPM


class DDPMModel(DDPM):
    def _pred_code(self):
        return {"eps": _lib.PRED_EPS, "x0": _lib.PRED_X0, "v": _lib.PRED_V}[self.parameterization]

    def _schedule(self, clip_denoised=True) -> Schedule:
        T = self.num_timesteps
        idx = np.arange(T - 1, -1, -1)
        g = lambda name: getattr(self, name).detach().cpu().numpy()[idx]      # fp32 buffers (extract_into_tensor)
        coef = np.zeros((T, _lib.DSD_NCOEF), dtype=np.float32)
        coef[:, 0], coef[:, 1] = g("sqrt_alphas_cumprod"), g("sqrt_one_minus_alphas_cumprod")
        coef[:, 2], coef[:, 3] = g("sqrt_recip_alphas_cumprod"), g("sqrt_recipm1_alphas_cumprod")
        coef[:, 4], coef[:, 5] = g("posterior_mean_coef1"), g("posterior_mean_coef2")
        coef[:, 6] = g("posterior_log_variance_clipped")
        return Schedule(_lib.MODE_B_DDPM, self._pred_code(), coef, idx.astype(np.float32), (idx != 0).astype(np.int32),
                        clip_denoised=clip_denoised)

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """:484-499."""
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            key = "c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"
            cond = {key: cond}
        x_recon = self.model(x_noisy, t, **cond)
        if isinstance(x_recon, tuple) and not return_ids:
            return x_recon[0]
        return x_recon

    @torch.no_grad()
    def p_sample_loop(self, shape, cond=None, return_intermediates=False, x_T=None, step_noise=None, seed=None,
                      *args, **kwargs):
        """:447-459.  ``x_T`` / ``step_noise`` / ``seed`` are extensions for reproducible runs."""
        device = self.betas.device
        img = x_T if x_T is not None else torch.randn(shape, device=device)
        c = cond["c_concat"] if isinstance(cond, dict) else (cond if isinstance(cond, list) else [cond])
        unet = find_unet(self.model)
        out = run_device_loop(unet, self._schedule(self.clip_denoised), img.to(device),
                              torch.cat([t.to(device) for t in c], 1), step_noise=step_noise, seed=seed)
        if return_intermediates:
            return out, [img, out]
        return out

    def sample(self, batch_size=16, cond=None, shape=None, return_intermediates=False, *args, **kwargs):
        """:442-445."""
        return self.p_sample_loop((batch_size, shape[0], shape[1], shape[2]), cond,
                                  return_intermediates=return_intermediates, **kwargs)

    @torch.no_grad()
    def log_images(self, batch, N=8, n_row=2, sample=True, return_keys=None, sampler="ddim", pred_mode=False,
                   ddim_eta=0, **kwargs):
        """:394-440 — the sampler switch of the prediction path (``dpm`` | ``ddim`` | DDPM ancestral).  ``batch`` holds
        the target under ``first_stage_key`` (shape only) and the condition under ``"image"``."""
        from ..ldm.models.diffusion.ddim import DDIMSampler
        from ..ldm.models.diffusion.dpm_solver_new import DPMSolverSampler
        device = self.betas.device
        x = batch[getattr(self, "first_stage_key", "t1ce")]
        c = batch["image"]
        N = min(x.shape[0], N)
        x, c = x.to(device)[:N], c.to(device)[:N]
        log = {"inputs": x}
        shape = x.shape[1:]
        for k in ("use_ema_scope",):
            kwargs.pop(k, None)
        if sample:
            if sampler == "dpm":
                kwargs.pop("ddim_use_original_steps", None)
                samples, _ = DPMSolverSampler(self).sample(kwargs.pop("ddim_steps"), N, shape, c, **kwargs)
            elif sampler == "ddim":
                samples, _ = DDIMSampler(self).sample(kwargs.pop("ddim_steps"), N, shape, c, verbose=False, eta=ddim_eta,
                                                      **kwargs)
            else:
                samples = self.sample(batch_size=N, cond=c, shape=shape)
            log["samples"] = samples
        if return_keys:
            keep = [k for k in log if k in return_keys]
            if keep:
                return {k: log[k] for k in keep}
        return log
