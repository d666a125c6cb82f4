"""DDIMSampler — drop-in for ldm/models/diffusion/ddim.py (make_schedule :25-55, sample :57-126,
ddim_sampling :128-185, p_sample_ddim :187-261), executed by dsd_sample (mode B_DDIM)."""
from __future__ import annotations

import numpy as np
import torch

from .... import _lib
from ...._sched import Schedule, find_unet, run_device_loop


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """ldm/modules/diffusionmodules/util.py:53-67 (+1 offset kept)."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, stride)))
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = steps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:70-81; ``alphacums`` = the model's fp32 alphas_cumprod buffer as a numpy array."""
    a_t = alphacums[ddim_timesteps]
    a_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))
    return sigmas, a_t, a_prev


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", device=torch.device("cuda"), **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.device = device

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        acp = self.model.alphas_cumprod.detach().float().cpu().numpy()
        assert acp.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        sig, a_t, a_prev = make_ddim_sampling_parameters(acp, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sigmas = np.asarray(sig, dtype=np.float64)
        self.ddim_alphas = np.asarray(a_t)
        self.ddim_alphas_prev = np.asarray(a_prev)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - self.ddim_alphas)          # fp32, like np.sqrt(1 - fp32 tensor)
        acp_prev = self.model.alphas_cumprod_prev.detach().float().cpu()
        acp_t = self.model.alphas_cumprod.detach().float().cpu()
        self.ddim_sigmas_for_original_num_steps = (ddim_eta * torch.sqrt(
            (1 - acp_prev) / (1 - acp_t) * (1 - acp_t / acp_prev))).numpy()

    def _schedule(self, use_original_steps: bool, clip_denoised: bool) -> Schedule:
        m = self.model
        buf = lambda name: getattr(m, name).detach().float().cpu().numpy()
        if use_original_steps:
            ts = np.arange(self.ddpm_num_timesteps)
            alphas, alphas_prev = buf("alphas_cumprod"), buf("alphas_cumprod_prev")
            s1m, sig = buf("sqrt_one_minus_alphas_cumprod"), self.ddim_sigmas_for_original_num_steps
        else:
            ts = self.ddim_timesteps
            alphas, alphas_prev = self.ddim_alphas, self.ddim_alphas_prev
            s1m, sig = self.ddim_sqrt_one_minus_alphas, self.ddim_sigmas
        n = len(ts)
        order = np.arange(n - 1, -1, -1)              # index = total_steps - i - 1  (:163)
        steps = np.asarray(ts)[order]                 # np.flip(timesteps)
        coef = np.zeros((n, _lib.DSD_NCOEF), dtype=np.float32)
        coef[:, 0] = buf("sqrt_alphas_cumprod")[steps]               # predict_*_from_z_and_v gather by t
        coef[:, 1] = buf("sqrt_one_minus_alphas_cumprod")[steps]
        coef[:, 4] = np.asarray(alphas)[order].astype(np.float32)    # torch.full(..., alphas[index]) -> fp32
        coef[:, 5] = np.asarray(alphas_prev)[order].astype(np.float32)
        coef[:, 6] = np.asarray(sig)[order].astype(np.float32)
        coef[:, 7] = np.asarray(s1m)[order].astype(np.float32)
        pred = {"eps": _lib.PRED_EPS, "v": _lib.PRED_V}[m.parameterization]
        return Schedule(_lib.MODE_B_DDIM, pred, coef, steps.astype(np.float32), np.ones(n, dtype=np.int32),
                        clip_denoised=clip_denoised)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               clip_denoised=True, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
               ucg_schedule=None, step_noise=None, seed=None, **kwargs):
        """:57-126 -> (samples, intermediates).  Unsupported reference options raise instead of being ignored."""
        if mask is not None or quantize_x0 or score_corrector is not None or dynamic_threshold is not None \
                or unconditional_guidance_scale != 1. or temperature != 1. or noise_dropout != 0. or ucg_schedule is not None:
            raise NotImplementedError("inpainting mask / guidance / quantisation options are not on the medical hot path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose,
                           ddim_discretize=kwargs.get("ddim_discretize", "uniform"))
        C_, H, W = shape
        size = (batch_size, C_, H, W)
        use_orig = kwargs.get("ddim_use_original_steps", False)
        device = self.model.betas.device
        img = x_T if x_T is not None else torch.randn(size, device=device)
        c = conditioning["c_concat"] if isinstance(conditioning, dict) else (
            conditioning if isinstance(conditioning, list) else [conditioning])
        unet = find_unet(self.model.model if hasattr(self.model, "model") else self.model)
        out = run_device_loop(unet, self._schedule(use_orig, clip_denoised), img.to(device),
                              torch.cat([t.to(device) for t in c], 1), step_noise=step_noise, seed=seed)
        return out, {"x_inter": [img, out], "pred_x0": [img, out]}
