"""GaussianDiffusion — drop-in for Disc_diff/guided_diffusion/gaussian_diffusion.py (sampling side).

Same constructor, enums and sampling entry points as the reference (``p_sample_loop`` :524-567,
``ddim_sample_loop`` :705-737, ``p_sample`` :422-465, ``ddim_sample`` :618-665 and their
``*_progressive`` generators).  The float64 schedule tables are built on the host exactly as
:141-178 does; the loop itself (network + fused update) runs on the MI355X through dsd_sample.
Training losses are out of scope.
"""
from __future__ import annotations

import enum
import math

import numpy as np
import torch as th

from ... import _lib
from ..._sched import Schedule, find_unet, run_device_loop, sampler_update, _seed_from_torch


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:31-54."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:57-73."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False,
                 parameterization="eps"):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        self.parameterization = parameterization
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        one_minus = 1.0 - betas
        self.alphas_cumprod = np.cumprod(one_minus, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(one_minus) / (1.0 - self.alphas_cumprod)

    # ------------------------------------------------------------------ host -> device schedule
    def _model_timestep_values(self) -> np.ndarray:
        """Value handed to the network for loop index i (``_scale_timesteps`` :381-384; overridden by SpacedDiffusion)."""
        t = np.arange(self.num_timesteps, dtype=np.float32)
        if self.rescale_timesteps:
            t = t * np.float32(1000.0 / self.num_timesteps)
        return t

    def _pred_code(self) -> int:
        if self.parameterization == "v":
            return _lib.PRED_V
        if self.model_mean_type == ModelMeanType.EPSILON:
            return _lib.PRED_EPS
        if self.model_mean_type == ModelMeanType.START_X:
            return _lib.PRED_X0
        raise NotImplementedError("ModelMeanType.PREVIOUS_X is not used by any shipped config")

    def _schedule(self, ddim: bool, eta: float, clip_denoised: bool) -> Schedule:
        T = self.num_timesteps
        idx = np.arange(T - 1, -1, -1)                          # loop order: i = T-1 .. 0  (:595,:764)
        f32 = lambda a: np.asarray(a, dtype=np.float64)[idx].astype(np.float32)   # arr[t].float()  (:1003)
        coef = np.zeros((T, _lib.DSD_NCOEF), dtype=np.float32)
        coef[:, 0] = f32(self.sqrt_alphas_cumprod)
        coef[:, 1] = f32(self.sqrt_one_minus_alphas_cumprod)
        coef[:, 2] = f32(self.sqrt_recip_alphas_cumprod)
        coef[:, 3] = f32(self.sqrt_recipm1_alphas_cumprod)
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE
        if self.model_var_type == ModelVarType.LEARNED:
            raise NotImplementedError("ModelVarType.LEARNED is not used by any shipped config")
        if ddim:
            coef[:, 4] = f32(self.alphas_cumprod)
            coef[:, 5] = f32(self.alphas_cumprod_prev)
        else:
            coef[:, 4] = f32(self.posterior_mean_coef1)
            coef[:, 5] = f32(self.posterior_mean_coef2)
            if learned:                                          # :287-290
                coef[:, 6] = f32(self.posterior_log_variance_clipped)
                coef[:, 7] = f32(np.log(self.betas))
            elif self.model_var_type == ModelVarType.FIXED_LARGE:   # :299-302
                coef[:, 6] = f32(np.log(np.append(self.posterior_variance[1], self.betas[1:])))
            else:                                                # FIXED_SMALL :303-306
                coef[:, 6] = f32(self.posterior_log_variance_clipped)
        t_model = self._model_timestep_values()[idx]
        nonzero = (idx != 0).astype(np.int32)                    # :457-459
        mode = _lib.MODE_A_DDIM if ddim else _lib.MODE_A_DDPM
        return Schedule(mode, self._pred_code(), coef, t_model, nonzero, learned_range=learned,
                        clip_denoised=clip_denoised, eta=eta)

    # ------------------------------------------------------------------ loops
    def _loop(self, ddim, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
              eta=0.0, step_noise=None, seed=None):
        assert isinstance(shape, (tuple, list))
        model_kwargs = model_kwargs or {}
        sched = self._schedule(ddim, eta, clip_denoised)
        hooks = denoised_fn is not None or cond_fn is not None
        self._check_hooks(ddim, denoised_fn, cond_fn)
        unet = find_unet(model)
        if hooks and unet is not None:
            # a Python hook sits between network and update: per-step host loop; a bare native U-Net does not concatenate
            # c_concat itself (DiffusionWrapper.forward does, ddpm.py:1328-1333)
            if unet is model:
                model = lambda x, t, _u=unet, **kw: _u(th.cat([x] + [c.to(x.device) for c in kw.get("c_concat", [])], 1), t)
            unet = None
        if device is None:
            device = next(model.parameters()).device if hasattr(model, "parameters") else th.device("cuda")
            if th.device(device).type != "cuda":
                device = th.device("cuda")
        img = noise if noise is not None else th.randn(*shape, device=device)      # :591-594
        img = img.to(device)
        c_concat = model_kwargs.get("c_concat")
        if unet is not None and c_concat is not None:
            cond = th.cat([c.to(device) for c in c_concat], 1)
            return run_device_loop(unet, sched, img, cond, step_noise=step_noise, seed=seed)
        # generic callable (closure / foreign module): python loop, fused HIP update per step
        seed = _seed_from_torch() if seed is None else seed
        x = img.float().contiguous().clone()
        B = x.shape[0]
        tvals = th.from_numpy(sched.t_model)
        for k in range(sched.steps):
            tv = tvals[k]
            is_int = (not self.rescale_timesteps) and float(tv) == int(tv)
            t = th.full((B,), int(tv) if is_int else float(tv), device=device,
                        dtype=th.long if is_int else th.float32)
            out = model(x, t, **model_kwargs)
            if isinstance(out, tuple):
                out = out[0]
            self._hooked_update(sched, k, out, x, t, None if step_noise is None else step_noise[k].to(device), seed,
                                denoised_fn, cond_fn, model_kwargs)
        return x

    # ---- denoised_fn / cond_fn (:312-313, :386-398, :460-463)
    def _check_hooks(self, ddim, denoised_fn, cond_fn):
        if cond_fn is None:
            return
        if ddim:
            raise NotImplementedError(
                "cond_fn with DDIM (condition_score, gaussian_diffusion.py:400-420) is not built: it re-derives pred_xstart from a "
                "shifted eps WITHOUT clipping it again, which the fused update cannot express.  Use p_sample_loop(..., cond_fn=...) "
                "(classifier guidance on the DDPM mean, supported for the fixed variance types), or write the step loop over "
                "model(x, t) and _sched.sampler_update() yourself")
        if self.model_var_type == ModelVarType.LEARNED_RANGE:
            raise NotImplementedError(
                "cond_fn needs the per-pixel variance of the step; with learn_sigma it lives inside the fused update.  Use a "
                "fixed-variance diffusion (learn_sigma=False) with p_sample_loop(..., cond_fn=...)")

    def _hooked_update(self, sched, k, out, x, t, noise, seed, denoised_fn, cond_fn, model_kwargs, want_x0=False):
        """One update with the reference's hooks, still on the fused HIP kernel.
        denoised_fn acts on the predicted x_start BEFORE the clip (process_xstart :311-316): x_start is formed here from the
        network output with the reference's fp32 expressions, handed to denoised_fn, and the kernel is then run in its
        x_start-prediction mode (it clips and does the rest).  cond_fn shifts the mean by variance * gradient (:386-398);
        the sample is linear in the mean, so the shift is added to the kernel's result."""
        if denoised_fn is None and cond_fn is None:
            return sampler_update(sched, k, out, x, noise, seed, want_x0=want_x0)
        grad = None
        if cond_fn is not None:
            grad = cond_fn(x, t, **model_kwargs).float()              # x is still x_t here; t as the network received it
        run = sched
        if denoised_fn is not None:
            c = th.from_numpy(sched.coef[k]).to(x.device)
            xs = x.shape
            C = xs[1]
            eps_or_v, rest = (out[:, :C], out[:, C:]) if sched.c.learned_range else (out, None)
            if sched.c.pred == _lib.PRED_V:                            # predict_start_from_z_and_v :236-242
                x0 = c[0] * x - c[1] * eps_or_v
            elif sched.c.pred == _lib.PRED_EPS:                        # _predict_xstart_from_eps :352-357
                x0 = c[2] * x - c[3] * eps_or_v
            else:
                x0 = eps_or_v
            x0 = denoised_fn(x0).float()
            out = x0 if rest is None else th.cat([x0, rest], 1)
            run = sched.with_pred(_lib.PRED_X0)
        x0_out = sampler_update(run, k, out, x, noise, seed, want_x0=want_x0)
        if grad is not None:
            # p_mean_var["variance"] of this step (:296-309): the variance TABLE (FIXED_SMALL pairs the raw posterior variance,
            # 0 at t = 0, with the clipped log), gathered in float64 and rounded to fp32 as _extract_into_tensor does
            i = self.num_timesteps - 1 - k
            var = (np.append(self.posterior_variance[1], self.betas[1:]) if self.model_var_type == ModelVarType.FIXED_LARGE
                   else self.posterior_variance)
            x.add_(float(np.float32(var[i])) * grad)
        return x0_out

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, step_noise=None, seed=None):
        """:524-567.  ``step_noise`` ([T,B,1,H,W]) / ``seed`` are extensions: injected normals (parity runs) or
        the Philox seed; by default the per-step noise is on-device Philox seeded from torch's generator."""
        return self._loop(False, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                          progress, 0.0, step_noise, seed)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, step_noise=None, seed=None):
        """:705-737."""
        return self._loop(True, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                          progress, eta, step_noise, seed)

    def dpm_solver_sample_loop(self, model, shape, clip_denoised=None, model_kwargs=None, noise=None, device=None):
        """:467-522 — DPM-Solver++ multistep order 2 over this (possibly respaced) schedule: ``num_timesteps`` network
        evaluations, logSNR spacing, dynamic thresholding, no lower-order final step.  As in the reference the network
        receives the solver's continuous model time (no timestep_map / rescale) and ``clip_denoised`` is unused.
        ``noise`` (x_T) is an extension for reproducible runs; the reference draws it itself (:470)."""
        from .sampler import NoiseScheduleVP, model_wrapper, DPM_Solver
        if device is None:
            device = next(model.parameters()).device if hasattr(model, "parameters") else th.device("cuda")
            if th.device(device).type != "cuda":
                device = th.device("cuda")
        x = (noise if noise is not None else th.randn(*shape, device=device)).to(device)
        thresholding, denoise = True, False
        noise_schedule = NoiseScheduleVP(schedule="discrete", betas=th.from_numpy(self.betas).float())
        model_fn_continuous = model_wrapper(model, noise_schedule, model_type="noise", model_kwargs=model_kwargs or {})
        dpm_solver = DPM_Solver(model_fn_continuous, noise_schedule, algorithm_type="dpmsolver++",
                                correcting_x0_fn="dynamic_thresholding" if thresholding else None)
        return dpm_solver.sample(x, steps=(self.num_timesteps - 1 if denoise else self.num_timesteps), order=2,
                                 skip_type="logSNR", method="multistep", lower_order_final=False,
                                 denoise_to_zero=denoise, solver_type="dpmsolver")

    def _progressive(self, ddim, model, shape, noise, clip_denoised, model_kwargs, device, eta, denoised_fn=None, cond_fn=None):
        model_kwargs = model_kwargs or {}
        sched = self._schedule(ddim, eta, clip_denoised)
        self._check_hooks(ddim, denoised_fn, cond_fn)
        if device is None:
            device = th.device("cuda")
        img = (noise if noise is not None else th.randn(*shape, device=device)).to(device).float().contiguous().clone()
        seed = _seed_from_torch()
        B = img.shape[0]
        for k in range(sched.steps):
            i = self.num_timesteps - 1 - k
            t = th.tensor([i] * B, device=device)
            out = self._call_model(model, img, t, model_kwargs)
            x0 = self._hooked_update(sched, k, out, img, self._model_t(t), None, seed, denoised_fn, cond_fn, model_kwargs, want_x0=True)
            yield {"sample": img.clone(), "pred_xstart": x0}

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """:569-616 (yields {"sample","pred_xstart"} per step)."""
        yield from self._progressive(False, model, shape, noise, clip_denoised, model_kwargs, device, 0.0, denoised_fn, cond_fn)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0):
        """:739-786."""
        yield from self._progressive(True, model, shape, noise, clip_denoised, model_kwargs, device, eta, denoised_fn, cond_fn)

    # ------------------------------------------------------------------ single steps
    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _model_t(self, t):
        """loop index -> the timestep the network (and cond_fn) receives (respace.py:123-128)."""
        tv = th.from_numpy(self._model_timestep_values()).to(t.device)[t]
        return tv if self.rescale_timesteps else tv.long()

    def _call_model(self, model, x, t, model_kwargs):
        """p_mean_variance's network call (:266-278): loop index -> network timestep, tuple -> tensor."""
        tv = th.from_numpy(self._model_timestep_values()).to(x.device)[t]
        if not self.rescale_timesteps:
            tv = tv.long()
        out = model(x, tv, **model_kwargs)
        return out[0] if isinstance(out, tuple) else out

    def _single(self, ddim, model, x, t, clip_denoised, model_kwargs, eta, denoised_fn=None, cond_fn=None):
        model_kwargs = model_kwargs or {}
        i = int(t[0])
        assert bool((t == i).all()), "all samples of a batch share the timestep in every reference loop"
        sched = self._schedule(ddim, eta, clip_denoised)
        self._check_hooks(ddim, denoised_fn, cond_fn)
        out = self._call_model(model, x, t, model_kwargs)
        xn = x.float().contiguous().clone()
        x0 = self._hooked_update(sched, self.num_timesteps - 1 - i, out, xn, self._model_t(t.to(x.device)), None, _seed_from_torch(),
                                 denoised_fn, cond_fn, model_kwargs, want_x0=True)
        return {"sample": xn, "pred_xstart": x0}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """:422-465."""
        return self._single(False, model, x, t, clip_denoised, model_kwargs, 0.0, denoised_fn, cond_fn)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """:618-665."""
        return self._single(True, model, x, t, clip_denoised, model_kwargs, eta, denoised_fn, cond_fn)


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """:994-1006."""
    res = th.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)
