"""DPMSolverSampler — drop-in for ldm/models/diffusion/dpm_solver_new/sampler.py:20-103: DPM-Solver++ multistep order 2,
uniform time spacing, over the LDM model's fp32 ``betas`` buffer; the loop runs in dsd_sample_dpm."""
from __future__ import annotations

import torch

from .dpm_solver_pytorch import NoiseScheduleVP, model_wrapper, DPM_Solver

MODEL_TYPES = {"eps": "noise", "v": "v"}


class _ApplyModel:
    """``lambda x, t, c: model.apply_model(x, t, c)`` of the reference (:90), kept as an object so the solver can find
    the native network behind it."""

    def __init__(self, ldm_model):
        self.ldm_model = ldm_model
        self.diffusion_model = getattr(getattr(ldm_model, "model", None), "diffusion_model", None)

    def __call__(self, x, t, c):
        return self.ldm_model.apply_model(x, t, c)


class DPMSolverSampler(object):
    def __init__(self, model, device=torch.device("cuda"), **kwargs):
        self.model = model
        self.device = device
        self.alphas_cumprod = model.alphas_cumprod.clone().detach().to(torch.float32)
        self.betas = model.betas.clone().detach().to(torch.float32)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        """:35-103 -> (samples, None)."""
        if mask is not None or quantize_x0 or score_corrector is not None or \
                (unconditional_guidance_scale != 1. and unconditional_conditioning is not None):
            raise NotImplementedError("inpainting mask / guidance / quantisation options are not on the medical hot path")
        C_, H, W = shape
        size = (batch_size, C_, H, W)
        device = self.model.betas.device
        img = torch.randn(size, device=device) if x_T is None else x_T
        ns = NoiseScheduleVP("discrete", betas=self.betas)
        if not isinstance(conditioning, dict):
            conditioning = dict(c_concat=conditioning if isinstance(conditioning, list) else [conditioning])
        model_fn = model_wrapper(_ApplyModel(self.model), ns, model_type=MODEL_TYPES[self.model.parameterization],
                                 guidance_type="classifier-free", condition=conditioning,
                                 unconditional_condition=unconditional_conditioning,
                                 guidance_scale=unconditional_guidance_scale)
        dpm_solver = DPM_Solver(model_fn, ns, algorithm_type="dpmsolver++")
        x = dpm_solver.sample(img.to(device), steps=S, skip_type="time_uniform", method="multistep", order=2)
        return x.to(device), None
