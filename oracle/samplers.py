"""Oracle (test infrastructure): CPU restatement of the reference sampling loops.

Family A: Disc_diff/guided_diffusion/gaussian_diffusion.py (p_mean_variance :244-350,
p_sample :422-465, ddim_sample :618-665, loops :569-616/:739-786) + respace.py:63-128.
Family B: trainers/trainer_ddpm.py:447-499 (+ ldm/models/diffusion/ddpm.py:290-311) and
ldm/models/diffusion/ddim.py:25-55,128-261.

``model(x_in, t) -> eps_or_v`` is a callable on CPU tensors; noise is injected (pre-drawn) so a
run is reproducible against the HIP path: ``noise[k]`` is the k-th ``randn_like`` the loop draws.
"""
from __future__ import annotations

import numpy as np
import torch

from . import schedules as S


def _ext(arr: np.ndarray, t: torch.Tensor, shape) -> torch.Tensor:
    """_extract_into_tensor gaussian_diffusion.py:994-1006: float64 gather, then .float()."""
    res = torch.from_numpy(np.asarray(arr))[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


class DiffusionA:
    """SpacedDiffusion(use_timesteps, betas, EPSILON|START_X, FIXED_*|LEARNED_RANGE, parameterization)."""

    def __init__(self, *, steps=1000, noise_schedule="linear", timestep_respacing="", predict_xstart=False,
                 sigma_small=False, learn_sigma=False, rescale_timesteps=False, parameterization="eps"):
        base = S.named_beta_schedule(noise_schedule, steps)               # script_util.py:142
        if not timestep_respacing:
            timestep_respacing = [steps]
        use = S.space_timesteps(steps, timestep_respacing)
        betas, self.timestep_map = S.spaced(base, use)
        self.original_num_steps = steps
        self.tab = S.gaussian_tables(betas)
        self.betas = self.tab["betas"]
        self.num_timesteps = len(betas)
        self.predict_xstart = predict_xstart
        self.var_type = "learned_range" if learn_sigma else ("fixed_small" if sigma_small else "fixed_large")
        self.rescale_timesteps = rescale_timesteps
        self.parameterization = parameterization

    # respace.py:123-128
    def model_t(self, t: torch.Tensor) -> torch.Tensor:
        new_ts = torch.tensor(self.timestep_map, dtype=t.dtype)[t]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return new_ts

    def p_mean_variance(self, model, x, t, clip_denoised=True, cond=None, denoised_fn=None):
        T = self.tab
        B, C = x.shape[:2]
        x_in = x if cond is None else torch.cat([x] + list(cond), 1)     # DiffusionWrapper ddpm.py:1331-1333
        out = model(x_in, self.model_t(t))
        if self.var_type == "learned_range":                               # :280-294
            out, var_values = torch.split(out, C, dim=1)
            min_log = _ext(T["posterior_log_variance_clipped"], t, x.shape)
            max_log = _ext(np.log(self.betas), t, x.shape)
            frac = (var_values + 1) / 2
            log_var = frac * max_log + (1 - frac) * min_log
        elif self.var_type == "fixed_large":                               # :299-302
            log_var = _ext(np.log(np.append(T["posterior_variance"][1], self.betas[1:])), t, x.shape)
        else:
            log_var = _ext(T["posterior_log_variance_clipped"], t, x.shape)
        def clip(z):                                                       # process_xstart :311-316
            if denoised_fn is not None:
                z = denoised_fn(z)
            return z.clamp(-1, 1) if clip_denoised else z
        if self.parameterization != "v":                                   # :318-333
            if self.predict_xstart:
                x0 = clip(out)
            else:
                x0 = clip(_ext(T["sqrt_recip_alphas_cumprod"], t, x.shape) * x
                          - _ext(T["sqrt_recipm1_alphas_cumprod"], t, x.shape) * out)
        else:                                                              # :336-338, :236-242
            x0 = clip(_ext(T["sqrt_alphas_cumprod"], t, x.shape) * x
                      - _ext(T["sqrt_one_minus_alphas_cumprod"], t, x.shape) * out)
        mean = (_ext(T["posterior_mean_coef1"], t, x.shape) * x0
                + _ext(T["posterior_mean_coef2"], t, x.shape) * x)         # :220-223
        return mean, log_var, x0

    @torch.no_grad()
    def p_sample_loop(self, model, x_T, noise, cond=None, clip_denoised=True, denoised_fn=None, cond_fn=None):
        """cond_fn(x, t_model, c_concat=...) -> gradient; condition_mean :386-398 through SpacedDiffusion's timestep wrapping
        (respace.py:94-100,123-128): mean += variance * gradient."""
        img = x_T
        for k, i in enumerate(reversed(range(self.num_timesteps))):
            t = torch.tensor([i] * x_T.shape[0])
            mean, log_var, _ = self.p_mean_variance(model, img, t, clip_denoised, cond, denoised_fn)
            if cond_fn is not None:
                grad = cond_fn(img, self.model_t(t), **({} if cond is None else dict(c_concat=list(cond))))
                # p_mean_var["variance"] (:296-309): the variance TABLE, not exp(log_variance) — FIXED_SMALL pairs the raw
                # posterior variance (0 at t = 0) with the clipped log
                var = (np.append(self.tab["posterior_variance"][1], self.betas[1:]) if self.var_type == "fixed_large"
                       else self.tab["posterior_variance"])
                assert self.var_type != "learned_range"
                mean = mean.float() + _ext(var, t, img.shape) * grad.float()
            nz = (t != 0).float().view(-1, 1, 1, 1)
            img = mean + nz * torch.exp(0.5 * log_var) * noise[k]          # :464
        return img

    @torch.no_grad()
    def ddim_sample_loop(self, model, x_T, noise, cond=None, clip_denoised=True, eta=0.0, denoised_fn=None):
        T = self.tab
        img = x_T
        for k, i in enumerate(reversed(range(self.num_timesteps))):
            t = torch.tensor([i] * x_T.shape[0])
            _, _, x0 = self.p_mean_variance(model, img, t, clip_denoised, cond, denoised_fn)
            eps = (_ext(T["sqrt_recip_alphas_cumprod"], t, img.shape) * img - x0) \
                / _ext(T["sqrt_recipm1_alphas_cumprod"], t, img.shape)     # :369-373
            ab = _ext(T["alphas_cumprod"], t, img.shape)
            abp = _ext(T["alphas_cumprod_prev"], t, img.shape)
            sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
            mean_pred = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
            nz = (t != 0).float().view(-1, 1, 1, 1)
            img = mean_pred + nz * sigma * noise[k]                        # :657-664
        return img


class DiffusionB:
    """DDPM schedule (ddpm.py:138-178, fp32 buffers) + DDPMModel.p_sample* (trainer_ddpm.py:447-482)."""

    def __init__(self, *, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 parameterization="v"):
        self.tab = {k: torch.from_numpy(v) for k, v in
                    S.ldm_tables(S.make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end)).items()}
        self.num_timesteps = timesteps
        self.parameterization = parameterization

    @staticmethod
    def _ext(a, t, shape):
        """extract_into_tensor util.py:103-106 (fp32 gather)."""
        return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(shape) - 1)))

    def x0_eps(self, x, t, out):
        T, e = self.tab, self._ext
        if self.parameterization == "v":                                   # ddpm.py:290-302
            x0 = e(T["sqrt_alphas_cumprod"], t, x.shape) * x - e(T["sqrt_one_minus_alphas_cumprod"], t, x.shape) * out
            eps = e(T["sqrt_alphas_cumprod"], t, x.shape) * out + e(T["sqrt_one_minus_alphas_cumprod"], t, x.shape) * x
        elif self.parameterization == "eps":                               # ddpm.py:284-288
            x0 = e(T["sqrt_recip_alphas_cumprod"], t, x.shape) * x - e(T["sqrt_recipm1_alphas_cumprod"], t, x.shape) * out
            eps = out
        else:
            x0, eps = out, None
        return x0, eps

    @torch.no_grad()
    def p_sample_loop(self, model, x_T, noise, cond=None, clip_denoised=True):
        T, e = self.tab, self._ext
        img = x_T
        b = x_T.shape[0]
        for k, i in enumerate(reversed(range(self.num_timesteps))):
            t = torch.full((b,), i, dtype=torch.long)
            x_in = img if cond is None else torch.cat([img] + list(cond), 1)
            x0, _ = self.x0_eps(img, t, model(x_in, t))
            if clip_denoised:
                x0 = x0.clamp(-1., 1.)
            mean = e(T["posterior_mean_coef1"], t, img.shape) * x0 + e(T["posterior_mean_coef2"], t, img.shape) * img
            lv = e(T["posterior_log_variance_clipped"], t, img.shape)
            nz = (1 - (t == 0).float()).reshape(b, 1, 1, 1)
            img = mean + nz * (0.5 * lv).exp() * noise[k]                  # trainer_ddpm.py:467
        return img

    @torch.no_grad()
    def ddim_sample(self, model, S_steps, x_T, noise, cond=None, eta=0.0, clip_denoised=True,
                    ddim_discretize="uniform", use_original_steps=False):
        """DDIMSampler.make_schedule/ddim_sampling/p_sample_ddim ddim.py:25-55,128-261."""
        ac = self.tab["alphas_cumprod"].numpy()
        if use_original_steps:                                            # ddim.py:52-55,145-153,230-233: index == t
            ts = np.arange(self.num_timesteps)
            acp = self.tab["alphas_cumprod_prev"]
            act = self.tab["alphas_cumprod"]
            sig = (eta * torch.sqrt((1 - acp) / (1 - act) * (1 - act / acp))).numpy()
            a, a_prev = ac, acp.numpy()
            sqrt_1ma = self.tab["sqrt_one_minus_alphas_cumprod"].numpy()
        else:
            ts = S.make_ddim_timesteps(ddim_discretize, S_steps, self.num_timesteps)
            sig, a, a_prev = S.make_ddim_sampling_parameters(ac, ts, eta)
            sqrt_1ma = np.sqrt(1. - a)
        img = x_T
        b = x_T.shape[0]
        total = ts.shape[0]
        for k, step in enumerate(np.flip(ts)):
            index = total - k - 1
            t = torch.full((b,), int(step), dtype=torch.long)
            x_in = img if cond is None else torch.cat([img] + list(cond), 1)
            out = model(x_in, t)
            x0, e_t = self.x0_eps(img, t, out)
            a_t = torch.full((b, 1, 1, 1), float(a[index]))
            a_p = torch.full((b, 1, 1, 1), float(a_prev[index]))
            s_t = torch.full((b, 1, 1, 1), float(sig[index]))
            s1 = torch.full((b, 1, 1, 1), float(sqrt_1ma[index]))
            if self.parameterization != "v":
                x0 = (img - s1 * e_t) / a_t.sqrt()
            if clip_denoised:
                x0 = x0.clamp(-1., 1.)
            dir_xt = (1. - a_p - s_t ** 2).sqrt() * e_t
            img = a_p.sqrt() * x0 + dir_xt + s_t * noise[k]
        return img
