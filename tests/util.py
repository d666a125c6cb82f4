"""Shared helpers for the tests (golden fixture access, synthetic tensors, error metrics)."""
import json
import os

import numpy as np
import torch

from oracle.synth import synth_params, randn, cond_image  # noqa: F401  (oracle = checker)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def fixture_params(g, key, prefix=""):
    """Regenerate the weights a fixture was produced with: stored (names, shapes, seed) only."""
    ns = [(n, tuple(s)) for n, s in json.loads(str(g[key + "_params"]))]
    sd = synth_params(ns, int(g[key + "_seed"]))
    return {prefix + k: v for k, v in sd.items()}


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64) if not torch.is_tensor(a) else a.double().cpu()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not torch.is_tensor(b) else b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---- DPM-Solver(++) cases of tests/golden/dpm.npz (tests/golden/gen_golden.py::gen_dpm) -------------------------------------
# key -> (betas source, model_type, keyword arguments of DPM_Solver(...).sample / oracle.dpm.dpm_multistep)
DPM_CASES = {
    "A_dpm_20": (("A", "linear", "20"), "noise", dict(steps=20, order=2, skip_type="logSNR", thresholding=True,
                                                       lower_order_final=False)),
    "A_dpm_cos_12": (("A", "cosine", "12"), "noise", dict(steps=12, order=2, skip_type="logSNR", thresholding=True,
                                                          lower_order_final=False)),
    "B_v_uniform_10": (("B", "betas"), "v", dict(steps=10, order=2, skip_type="time_uniform")),
    "B_v_uniform_6_lof": (("B", "betas"), "v", dict(steps=6, order=2, skip_type="time_uniform")),
    "B_eps_uniform_6": (("B", "betas"), "noise", dict(steps=6, order=2, skip_type="time_uniform")),
    "B_eps_acp_quad_7_thr": (("B", "acp"), "noise", dict(steps=7, order=2, skip_type="time_quadratic", thresholding=True,
                                                        lower_order_final=False)),
    "B_x0_order1_5": (("B", "betas"), "x_start", dict(steps=5, order=1, skip_type="time_uniform")),
    "B_eps_dz_6": (("B", "betas"), "noise", dict(steps=6, order=2, skip_type="logSNR", thresholding=True,
                                                 denoise_to_zero=True)),
    "B_eps_plain_8": (("B", "betas"), "noise", dict(steps=8, order=2, skip_type="time_uniform", algorithm="dpmsolver")),
    "B_v_taylor_8": (("B", "betas"), "v", dict(steps=8, order=2, skip_type="time_uniform", solver_type="taylor")),
    "B_eps_plain_taylor_range_6": (("B", "betas"), "noise", dict(steps=6, order=2, skip_type="logSNR", algorithm="dpmsolver",
                                                                solver_type="taylor", t_start=0.8, t_end=0.02,
                                                                lower_order_final=False)),
}


def dpm_case_betas(src):
    """fp32 betas / alphas_cumprod tensors exactly as the fixture generator handed them to NoiseScheduleVP."""
    from oracle import schedules as S
    if src[0] == "A":
        base = S.named_beta_schedule(src[1], 1000)
        betas, _ = S.spaced(base, S.space_timesteps(1000, src[2]))
        return dict(betas=torch.from_numpy(np.asarray(betas)).float())
    b64 = S.make_beta_schedule("linear", 1000, 1e-4, 2e-2)
    if src[1] == "betas":
        return dict(betas=torch.tensor(b64, dtype=torch.float32))
    return dict(alphas_cumprod=torch.tensor(np.cumprod(1. - b64, axis=0), dtype=torch.float32))
