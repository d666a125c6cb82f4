"""Oracle (test infrastructure): CPU restatement of the reference's DPM-Solver(++) multistep sampler.

Follows Disc_diff/guided_diffusion/sampler.py (NoiseScheduleVP 'discrete' :76-149, model_wrapper :236-279,
DPM_Solver.dynamic_thresholding_fn :379-388, data_prediction_fn :396-405, get_time_steps :416-443,
dpm_solver_first_update :509-553, multistep_dpm_solver_second_update :760-816, sample(method='multistep') :1130-1176,
interpolate_fn :1224-1262) and its twin ldm/models/diffusion/dpm_solver_new/dpm_solver_pytorch.py (same arithmetic;
the 'v'/'x_start' model types broadcast alpha/sigma per sample :283-299).  Call sites restated:
Disc_diff/guided_diffusion/gaussian_diffusion.py:467-522 (logSNR spacing, order 2, thresholding, no lower-order final)
and ldm/models/diffusion/dpm_solver_new/sampler.py:86-101 (time_uniform, order 2, lower_order_final).

All schedule scalars are fp32 torch CPU tensors, as in a CPU run of the reference.  Only tests, smoke() and the bench's
cpu_baseline may import this module.
"""
from __future__ import annotations

import torch


def pl_interp(x: torch.Tensor, xp: torch.Tensor, yp: torch.Tensor) -> torch.Tensor:
    """Piecewise-linear y(x) through (xp, yp) (xp ascending), end segments extended (sampler.py:1224-1262).

    The reference locates x by sorting [x, xp...] and taking the slot x lands in; where x equals a knot the slot
    depends on torch.sort's tie order, so the same primitive is used here to stay bit-identical on CPU."""
    K = xp.numel()
    out = torch.empty_like(x)
    for n in range(x.numel()):
        order = torch.sort(torch.cat([x[n:n + 1], xp]))[1]
        slot = int(torch.argmin(order))                      # number of knots sorted before x
        j = min(max(slot - 1, 0), K - 2)
        out[n] = yp[j] + (x[n] - xp[j]) * (yp[j + 1] - yp[j]) / (xp[j + 1] - xp[j])
    return out


class NoiseSchedule:
    """NoiseScheduleVP(schedule='discrete', betas=... | alphas_cumprod=...), sampler.py:76-149."""

    def __init__(self, betas=None, alphas_cumprod=None):
        if betas is not None:
            la = 0.5 * torch.log(1 - betas).cumsum(dim=0)
        else:
            la = 0.5 * torch.log(alphas_cumprod)
        # numerical_clip_alpha :93-104 — drop the tail whose half-logSNR falls below -5.1
        lam = la - 0.5 * torch.log(1. - torch.exp(2. * la))
        cut = int(torch.searchsorted(torch.flip(lam, [0]), torch.tensor(-5.1, dtype=lam.dtype)))
        if cut > 0:
            la = la[:-cut]
        self.log_alpha = la.to(torch.float32)
        self.total_N = la.numel()
        self.t_knots = torch.linspace(0., 1., self.total_N + 1)[1:].to(torch.float32)
        self.T = 1.

    def log_mean(self, t):                                   # marginal_log_mean_coeff :106-114
        return pl_interp(t.reshape(-1), self.t_knots, self.log_alpha)

    def alpha(self, t):
        return torch.exp(self.log_mean(t))

    def std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_mean(t)))

    def lam(self, t):                                        # marginal_lambda :128-134
        lm = self.log_mean(t)
        return lm - 0.5 * torch.log(1. - torch.exp(2. * lm))

    def inv_lam(self, lamb):                                 # inverse_lambda :136-148
        la = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return pl_interp(la.reshape(-1), torch.flip(self.log_alpha, [0]), torch.flip(self.t_knots, [0]))


def time_steps(ns: NoiseSchedule, skip_type: str, t_T: float, t_0: float, N: int) -> torch.Tensor:
    """get_time_steps :416-443."""
    if skip_type == "logSNR":
        lT = ns.lam(torch.tensor(t_T))
        l0 = ns.lam(torch.tensor(t_0))
        return ns.inv_lam(torch.linspace(lT.item(), l0.item(), N + 1))
    if skip_type == "time_uniform":
        return torch.linspace(t_T, t_0, N + 1)
    if skip_type == "time_quadratic":
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
    raise ValueError(skip_type)


def dynamic_threshold(x0, ratio=0.995, max_val=1.):
    """dynamic_thresholding_fn :379-388."""
    s = torch.quantile(torch.abs(x0).reshape((x0.shape[0], -1)), ratio, dim=1)
    s = torch.maximum(s, max_val * torch.ones_like(s)).reshape(-1, *([1] * (x0.dim() - 1)))
    return torch.clamp(x0, -s, s) / s


@torch.no_grad()
def dpm_multistep(model, ns: NoiseSchedule, x, *, steps, order=2, skip_type="time_uniform", model_type="noise",
                  algorithm="dpmsolver++", thresholding=False, lower_order_final=True, denoise_to_zero=False,
                  solver_type="dpmsolver", t_start=None, t_end=None, ratio=0.995, max_val=1.):
    """DPM_Solver.sample(method='multistep') :1130-1176.  ``model(x, t_input[B] fp32) -> out[B,C,H,W]``."""
    assert order in (1, 2) and steps >= order
    t_0 = 1. / ns.total_N if t_end is None else t_end
    t_T = ns.T if t_start is None else t_start
    ts = time_steps(ns, skip_type, t_T, t_0, steps)
    B = x.shape[0]

    def noise_pred(x, t):                                    # model_wrapper.noise_pred_fn :247-265
        tc = t.expand(B)
        out = model(x, (tc - 1. / ns.total_N) * 1000.)
        if model_type == "noise":
            return out
        a, s = ns.alpha(tc)[:, None, None, None], ns.std(tc)[:, None, None, None]
        if model_type == "x_start":
            return (x - a * out) / s
        return a * out + s * x                               # "v"

    def model_fn(x, t):                                      # :390-414
        eps = noise_pred(x, t)
        if algorithm != "dpmsolver++":
            return eps
        x0 = (x - ns.std(t) * eps) / ns.alpha(t)
        return dynamic_threshold(x0, ratio, max_val) if thresholding else x0

    def first(x, s, t, m):                                   # :509-553
        h = ns.lam(t) - ns.lam(s)
        if algorithm == "dpmsolver++":
            return ns.std(t) / ns.std(s) * x - ns.alpha(t) * torch.expm1(-h) * m
        return torch.exp(ns.log_mean(t) - ns.log_mean(s)) * x - (ns.std(t) * torch.expm1(h)) * m

    def second(x, m1, m0, t1, t0, t):                        # :760-816 (m1 older, m0 newest)
        l1, l0, lt = ns.lam(t1), ns.lam(t0), ns.lam(t)
        h0, h = l0 - l1, lt - l0
        D1 = (1. / (h0 / h)) * (m0 - m1)
        if algorithm == "dpmsolver++":
            p = torch.expm1(-h)
            a_t = torch.exp(ns.log_mean(t))
            base = (ns.std(t) / ns.std(t0)) * x - (a_t * p) * m0
            if solver_type == "dpmsolver":
                return base - 0.5 * (a_t * p) * D1
            return base + (a_t * (p / h + 1.)) * D1
        p = torch.expm1(h)
        s_t = ns.std(t)
        base = (torch.exp(ns.log_mean(t) - ns.log_mean(t0))) * x - (s_t * p) * m0
        if solver_type == "dpmsolver":
            return base - 0.5 * (s_t * p) * D1
        return base - (s_t * (p / h - 1.)) * D1

    t_prev = [ts[0]]
    m_prev = [model_fn(x, ts[0])]
    for step in range(1, order):
        x = first(x, t_prev[-1], ts[step], m_prev[-1])
        t_prev.append(ts[step])
        m_prev.append(model_fn(x, ts[step]))
    for step in range(order, steps + 1):
        t = ts[step]
        so = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
        if so == 1:
            x = first(x, t_prev[-1], t, m_prev[-1])
        else:
            x = second(x, m_prev[-2], m_prev[-1], t_prev[-2], t_prev[-1], t)
        for i in range(order - 1):
            t_prev[i], m_prev[i] = t_prev[i + 1], m_prev[i + 1]
        t_prev[-1] = t
        if step < steps:
            m_prev[-1] = model_fn(x, t)
    if denoise_to_zero:                                      # denoise_to_zero_fn :503-507
        t = torch.ones((1,)) * t_0
        eps = noise_pred(x, t)
        x0 = (x - ns.std(t) * eps) / ns.alpha(t)
        x = dynamic_threshold(x0, ratio, max_val) if thresholding else x0
    return x
