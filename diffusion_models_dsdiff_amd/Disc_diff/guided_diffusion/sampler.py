"""DPM-Solver / DPM-Solver++ multistep sampler — native counterpart of the reference's
Disc_diff/guided_diffusion/sampler.py (NoiseScheduleVP :7-149, model_wrapper :151-302, DPM_Solver :305-1222) and of
its twin ldm/models/diffusion/dpm_solver_new/dpm_solver_pytorch.py.

The noise-schedule scalars are host-side fp32 torch expressions in the reference's operation order; the sampling loop
(network evaluation, data/noise prediction, dynamic thresholding, first/second-order multistep update) runs on the
MI355X through ``dsd_sample_dpm`` (include/dsdiff.h).  Built: ``method='multistep'`` with ``order`` 1 or 2, both
``algorithm_type``s, both ``solver_type``s, all three ``skip_type``s, ``lower_order_final``, ``denoise_to_zero``,
dynamic thresholding.  Singlestep / adaptive solvers, order 3, guidance and python correctors raise
``NotImplementedError`` — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from ..._lib import DSD_NCOEF, DsdDpmSchedule, check, dptr, lib, stream_ptr
from ..._sched import find_unet

_PRED = {"noise": 0, "x_start": 1, "v": 2}


def interpolate_fn(x: torch.Tensor, xp: torch.Tensor, yp: torch.Tensor) -> torch.Tensor:
    """y(x) on the polyline (xp, yp); x [N,1], xp/yp [1,K]; outside the knots the end segments continue (:1224-1262).
    The knot interval is the slot x takes when sorted in with the knots (ties as torch.sort leaves them)."""
    K = xp.shape[1]
    merged = torch.cat([x, xp.expand(x.shape[0], K)], dim=1)
    slot = torch.argmin(torch.sort(merged, dim=1)[1], dim=1)             # knots sorted ahead of x
    j = (slot - 1).clamp(0, K - 2)
    x0, x1 = xp[0][j], xp[0][j + 1]
    y0, y1 = yp[0][j], yp[0][j + 1]
    return (y0 + (x[:, 0] - x0) * (y1 - y0) / (x1 - x0)).reshape(-1, 1)


class NoiseScheduleVP:
    """Forward VP-SDE wrapper (:7-149): log alpha_t, sigma_t, lambda_t = log alpha_t - log sigma_t and its inverse."""

    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20., dtype=torch.float32):
        if schedule not in ["discrete", "linear"]:
            raise ValueError(
                "Unsupported noise schedule {}. The schedule needs to be 'discrete' or 'linear'".format(schedule))
        self.schedule = schedule
        self.T = 1.
        if schedule == "discrete":
            if betas is not None:
                log_alphas = 0.5 * torch.log(1 - betas.detach().cpu()).cumsum(dim=0)
            else:
                assert alphas_cumprod is not None
                log_alphas = 0.5 * torch.log(alphas_cumprod.detach().cpu())
            self.log_alpha_array = self.numerical_clip_alpha(log_alphas).reshape((1, -1,)).to(dtype=dtype)
            self.total_N = self.log_alpha_array.shape[1]
            self.t_array = torch.linspace(0., 1., self.total_N + 1)[1:].reshape((1, -1)).to(dtype=dtype)
        else:
            self.total_N = 1000
            self.beta_0 = continuous_beta_0
            self.beta_1 = continuous_beta_1

    def numerical_clip_alpha(self, log_alphas, clipped_lambda=-5.1):
        """:93-104 — cut the tail of the schedule whose half-logSNR is below ``clipped_lambda``."""
        lambs = log_alphas - 0.5 * torch.log(1. - torch.exp(2. * log_alphas))
        idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(clipped_lambda, dtype=lambs.dtype)))
        return log_alphas[:-idx] if idx > 0 else log_alphas

    def marginal_log_mean_coeff(self, t):
        t = torch.as_tensor(t, dtype=torch.float32).cpu()
        if self.schedule == "discrete":
            return interpolate_fn(t.reshape((-1, 1)), self.t_array, self.log_alpha_array).reshape((-1))
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        log_mean_coeff = self.marginal_log_mean_coeff(t)
        return log_mean_coeff - 0.5 * torch.log(1. - torch.exp(2. * log_mean_coeff))

    def inverse_lambda(self, lamb):
        lamb = torch.as_tensor(lamb, dtype=torch.float32).cpu()
        if self.schedule == "linear":
            tmp = 2. * (self.beta_1 - self.beta_0) * torch.logaddexp(-2. * lamb, torch.zeros((1,)))
            Delta = self.beta_0 ** 2 + tmp
            return tmp / (torch.sqrt(Delta) + self.beta_0) / (self.beta_1 - self.beta_0)
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return interpolate_fn(log_alpha.reshape((-1, 1)), torch.flip(self.log_alpha_array, [1]),
                              torch.flip(self.t_array, [1])).reshape((-1,))


class _ModelFn:
    """What ``model_wrapper`` returns: the network, its output type and its conditioning, kept apart so the solver can
    hand them to the device loop.  Calling it evaluates the noise prediction at continuous time (:247-279)."""

    def __init__(self, model, noise_schedule, model_type, model_kwargs, condition):
        self.model, self.noise_schedule, self.model_type = model, noise_schedule, model_type
        self.model_kwargs, self.condition = dict(model_kwargs or {}), condition

    def input_time(self, t_continuous):
        """get_model_input_time :236-245."""
        if self.noise_schedule.schedule == "discrete":
            return (t_continuous - 1. / self.noise_schedule.total_N) * 1000.
        return t_continuous

    def network(self, x, t_input):
        if self.condition is None:
            out = self.model(x, t_input, **self.model_kwargs)
        else:
            out = self.model(x, t_input, self.condition, **self.model_kwargs)
        if isinstance(out, tuple):
            out = out[0]
        return out

    def c_concat(self) -> Optional[list]:
        src = self.condition if self.condition is not None else self.model_kwargs
        if isinstance(src, dict):
            src = src.get("c_concat")
        if isinstance(src, torch.Tensor):
            src = [src]
        return src

    def __call__(self, x, t_continuous):
        ns = self.noise_schedule
        t = torch.as_tensor(t_continuous, dtype=torch.float32).reshape(-1).expand(x.shape[0])
        out = self.network(x, self.input_time(t).to(x.device))
        if out.shape[1] == 2 * x.shape[1]:
            out = out[:, :x.shape[1]]
        if self.model_type == "noise":
            return out
        a = ns.marginal_alpha(t).to(x.device)[:, None, None, None]
        s = ns.marginal_std(t).to(x.device)[:, None, None, None]
        if self.model_type == "x_start":
            return (x - a * out) / s
        if self.model_type == "v":
            return a * out + s * x
        return -s * out                                                    # "score"


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", condition=None,
                  unconditional_condition=None, guidance_scale=1., classifier_fn=None, classifier_kwargs={}):
    """:151-302 / dpm_solver_pytorch.py:188-336.  ``model`` is the denoiser as the reference passes it (the native
    DSUnetModel, a DiffusionWrapper around it, or any callable ``model(x, t_input, [cond], **model_kwargs)``)."""
    assert model_type in ["noise", "x_start", "v", "score"]
    assert guidance_type in ["uncond", "classifier", "classifier-free"]
    if guidance_type == "classifier" or (guidance_type == "classifier-free" and guidance_scale != 1.
                                          and unconditional_condition is not None):
        raise NotImplementedError("classifier / classifier-free guidance is not part of the conditional-DDPM path")
    return _ModelFn(model, noise_schedule, model_type, model_kwargs, condition if guidance_type != "uncond" else None)


class DpmSchedule:
    """Owns the host arrays a dsd_dpm_schedule points to."""

    def __init__(self, pred, data_pred, thresholding, ratio, max_val, coef, t_input, order):
        steps = len(order)
        self.coef = np.ascontiguousarray(coef, dtype=np.float32).reshape(steps, DSD_NCOEF)
        self.t_input = np.ascontiguousarray(t_input, dtype=np.float32).reshape(steps)
        self.order = np.ascontiguousarray(order, dtype=np.int32).reshape(steps)
        self.c = DsdDpmSchedule()
        self.c.steps, self.c.pred, self.c.data_pred = steps, int(pred), int(data_pred)
        self.c.thresholding, self.c.threshold_ratio, self.c.threshold_max = int(thresholding), float(ratio), float(max_val)
        self.c.coef = self.coef.ctypes.data_as(C.POINTER(C.c_float))
        self.c.t_input = self.t_input.ctypes.data_as(C.POINTER(C.c_float))
        self.c.order = self.order.ctypes.data_as(C.POINTER(C.c_int32))

    @property
    def steps(self) -> int:
        return int(self.c.steps)


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1., dynamic_thresholding_ratio=0.995):
        """:306-377."""
        assert algorithm_type in ["dpmsolver", "dpmsolver++"]
        if not isinstance(model_fn, _ModelFn):
            raise TypeError("model_fn must come from this module's model_wrapper(...) so the solver can reach the network")
        if model_fn.model_type == "score":
            raise NotImplementedError("score-type networks are not part of the conditional-DDPM path")
        if correcting_xt_fn is not None or not (correcting_x0_fn is None or correcting_x0_fn == "dynamic_thresholding"):
            raise NotImplementedError("python corrector callbacks cannot run inside the device loop")
        self.model_fn_ = model_fn
        self.noise_schedule = noise_schedule
        self.algorithm_type = algorithm_type
        self.correcting_x0_fn = correcting_x0_fn
        self.correcting_xt_fn = None
        self.dynamic_thresholding_ratio = dynamic_thresholding_ratio
        self.thresholding_max_val = thresholding_max_val

    # ------------------------------------------------------------------ host tables
    def get_time_steps(self, skip_type, t_T, t_0, N, device=None):
        """:416-443 (N+1 times from t_T down to t_0)."""
        ns = self.noise_schedule
        if skip_type == "logSNR":
            lambda_T = ns.marginal_lambda(torch.tensor(t_T))
            lambda_0 = ns.marginal_lambda(torch.tensor(t_0))
            return ns.inverse_lambda(torch.linspace(lambda_T.item(), lambda_0.item(), N + 1))
        elif skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        elif skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError(
            "Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))

    def build_schedule(self, steps, t_start=None, t_end=None, order=2, skip_type="time_uniform",
                       lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver") -> DpmSchedule:
        """Per-evaluation coefficient rows of the multistep loop (:1130-1176) for dsd_sample_dpm."""
        if solver_type not in ["dpmsolver", "taylor"]:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        if order not in (1, 2):
            raise NotImplementedError("the device loop is built for multistep order 1 and 2 (got {})".format(order))
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        t_0 = 1. / ns.total_N if t_end is None else t_end
        t_T = ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in [1 / N, 1], where N is the length of betas array"
        assert steps >= order
        ts = self.get_time_steps(skip_type, t_T, t_0, steps)
        assert ts.shape[0] - 1 == steps
        rows, orders, tin = [], [], []
        f = lambda v: float(v.reshape(-1)[0])

        def row(t_eval, cx=0., cm=0., cd=0., ir0=0.):
            r = [f(ns.marginal_alpha(t_eval)), f(ns.marginal_std(t_eval)), cx, cm, cd, ir0] + [0.] * (DSD_NCOEF - 6)
            rows.append(r)
            tin.append(f(self.model_fn_.input_time(torch.as_tensor(t_eval, dtype=torch.float32).reshape(-1))))

        for k in range(steps):
            s, t, step = ts[k], ts[k + 1], k + 1
            if step < order:
                so = step
            elif lower_order_final and steps < 10:
                so = min(order, steps + 1 - step)
            else:
                so = order
            lam_s, lam_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
            h = lam_t - lam_s
            la_s, la_t = ns.marginal_log_mean_coeff(s), ns.marginal_log_mean_coeff(t)
            sig_s, sig_t = ns.marginal_std(s), ns.marginal_std(t)
            alpha_t = torch.exp(la_t)
            if pp:
                phi_1 = torch.expm1(-h)
                cx, lead = sig_t / sig_s, alpha_t
            else:
                phi_1 = torch.expm1(h)
                cx, lead = torch.exp(la_t - la_s), sig_t
            cm = lead * phi_1
            cd = ir0 = torch.zeros(1)
            if so == 2:
                h_0 = lam_s - ns.marginal_lambda(ts[k - 1])
                ir0 = 1. / (h_0 / h)
                if solver_type == "dpmsolver":
                    cd = 0.5 * (lead * phi_1)
                elif pp:
                    cd = -(lead * (phi_1 / h + 1.))
                else:
                    cd = lead * (phi_1 / h - 1.)
            row(s, f(cx), f(cm), f(cd), f(ir0))
            orders.append(so)
        if denoise_to_zero:
            row(torch.ones((1,)) * t_0)
            orders.append(0)
        thr = self.correcting_x0_fn == "dynamic_thresholding"
        return DpmSchedule(_PRED[self.model_fn_.model_type], pp, thr, self.dynamic_thresholding_ratio,
                           self.thresholding_max_val, rows, tin, orders)

    # ------------------------------------------------------------------ device loop
    @torch.no_grad()
    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               return_intermediate=False):
        """:1017-1222, ``method='multistep'``.  x: [B,1,H,W] x_T on the GPU; returns the sample at t_end."""
        if method != "multistep":
            raise NotImplementedError("only the multistep solver (the one the reference's call sites use) is built; got "
                                      + repr(method))
        if return_intermediate:
            raise NotImplementedError("intermediates are not kept by the device loop")
        sched = self.build_schedule(steps, t_start, t_end, order, skip_type, lower_order_final, denoise_to_zero,
                                    solver_type)
        return run_dpm_loop(self.model_fn_, sched, x)


@torch.no_grad()
def run_dpm_loop(fn: _ModelFn, sched: DpmSchedule, x_T: torch.Tensor) -> torch.Tensor:
    if not x_T.is_cuda:
        raise RuntimeError("sampling runs on the MI355X only (no CPU fallback): x is on the CPU")
    x = x_T.detach().float().contiguous().clone()
    B, Cx, H, W = x.shape
    assert Cx == 1, "the denoised image has one channel"
    unet, cc = find_unet(fn.model), fn.c_concat()
    if unet is not None and cc is not None:
        unet.sync_params()
        cond = torch.cat([c.to(x.device) for c in cc], 1).detach().float().contiguous()
        assert cond.shape[0] == B and cond.shape[2:] == x.shape[2:]
        check(lib().dsd_sample_dpm(unet._h, C.byref(sched.c), dptr(cond), cond.shape[1], dptr(x), B, H, W, stream_ptr()))
        return x
    # any other callable: python loop over the network, fused HIP post-network step per evaluation
    m_cur, m_prev = torch.empty_like(x), torch.empty_like(x)
    for k in range(sched.steps):
        t_in = torch.full((B,), float(sched.t_input[k]), device=x.device, dtype=torch.float32)
        out = fn.network(x, t_in).float().contiguous()
        check(lib().dsd_op_dpm_step(C.byref(sched.c), k, dptr(out), out.shape[1], dptr(x), dptr(m_cur), dptr(m_prev), B, H, W,
                                    stream_ptr()))
        m_cur, m_prev = m_prev, m_cur
    return x


@torch.no_grad()
def dynamic_thresholding(x0: torch.Tensor, ratio: float = 0.995, max_val: float = 1.):
    """DPM_Solver.dynamic_thresholding_fn :379-388 on the device.  Returns (thresholded x0, s [B])."""
    x0 = x0.detach().float().contiguous()
    B = x0.shape[0]
    y, s = torch.empty_like(x0), torch.empty(B, device=x0.device, dtype=torch.float32)
    check(lib().dsd_op_dpm_threshold(dptr(x0), B, x0[0].numel(), float(ratio), float(max_val), dptr(y), dptr(s), stream_ptr()))
    return y, s


def expand_dims(v, dims):
    """:1265-1274."""
    return v[(...,) + (None,) * (dims - 1)]
