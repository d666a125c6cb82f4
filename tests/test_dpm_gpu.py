"""DPM-Solver(++) multistep parity (GPU): dsd_sample_dpm / dsd_op_dpm_step / dsd_op_dpm_threshold through the reference's
call signatures vs fixtures produced by the reference's own solver (tests/golden/dpm.npz, tests/golden/gen_golden.py::gen_dpm)
and vs the oracle (oracle/dpm.py).  Selection (quantile) and the post-network arithmetic are bit-exact; whole sampled
images carry the network's fp32 tolerance, rel-L2 <= 1e-4 (BASELINE.json north_star), measured ~1e-6."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from oracle import dpm as ODPM
from util import DPM_CASES, dpm_case_betas, golden, fixture_params, rel_l2, randn, cond_image

pytestmark = pytest.mark.gpu
TOL = 1e-4
SHAPE = (2, 1, 32, 32)


@pytest.fixture(scope="module")
def env():
    from diffusion_models_dsdiff_amd import _lib
    from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddpm import DiffusionWrapper
    _lib.require_gpu(0)
    gm, g = golden("model"), golden("dpm")
    params = json.loads(str(gm["tiny_cfg"]))
    wrap = DiffusionWrapper({"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}, "concat")
    wrap.diffusion_model.load_state_dict(fixture_params(gm, "tiny"), strict=True)
    cond = cond_image(SHAPE, int(g["cond_seed"])).cuda()
    xT = randn(SHAPE, int(g["xT_seed"])).cuda()
    return g, wrap, cond, xT


def _solver(key, model, **wrap_kw):
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion import sampler as dsa
    src, mtype, kw = DPM_CASES[key]
    kw = dict(kw)
    ns = dsa.NoiseScheduleVP("discrete", **dpm_case_betas(src))
    fn = dsa.model_wrapper(model, ns, model_type=mtype, **wrap_kw)
    sol = dsa.DPM_Solver(fn, ns, algorithm_type=kw.pop("algorithm", "dpmsolver++"),
                         correcting_x0_fn="dynamic_thresholding" if kw.pop("thresholding", False) else None)
    return sol, kw


def test_dynamic_thresholding_bit_exact(env):
    """torch.quantile(|x0|, 0.995) per sample by radix select + clamp/divide: equal to the reference's CPU result bit for
    bit (fixtures), incl. a ragged size (17x23) and a case where the 1.0 floor wins."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.sampler import dynamic_thresholding
    g = env[0]
    for i in range(3):
        shp = tuple(int(v) for v in g[f"thr{i}_shape"])
        x0 = randn(shp, 90 + i) * float(g[f"thr{i}_scale"])
        y, s = dynamic_thresholding(x0.cuda())
        np.testing.assert_array_equal(s.cpu().numpy(), np.maximum(g[f"thr{i}_s"], np.float32(1.0)))
        np.testing.assert_array_equal(y.cpu().numpy(), g[f"thr{i}_y"])


def test_dynamic_thresholding_properties_full_size():
    """BASELINE slice size (256x256 = 65536 values per sample, batch 16) and 512x512; ties, constant input, single
    element, other ratios — against torch.quantile on the CPU, bit-exact."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.sampler import dynamic_thresholding
    gen = torch.Generator().manual_seed(7)
    cases = [torch.randn(16, 1, 256, 256, generator=gen) * 3.0,
             torch.randn(2, 1, 512, 512, generator=gen) * 0.2,
             torch.randint(-3, 4, (3, 1, 64, 64), generator=gen).float(),          # heavy ties
             torch.full((2, 1, 16, 16), -2.5),                                      # constant
             torch.randn(4, 1, 1, 1, generator=gen) * 5,                            # n = 1
             torch.randn(3, 1, 7, 3, generator=gen) * 5]
    for x0 in cases:
        for ratio in (0.995, 0.5, 1.0, 0.0):
            y, s = dynamic_thresholding(x0.cuda(), ratio, 1.0)
            want = ODPM.dynamic_threshold(x0, ratio, 1.0)
            q = torch.quantile(x0.abs().reshape(x0.shape[0], -1), ratio, dim=1).clamp_min(1.0)
            np.testing.assert_array_equal(s.cpu().numpy(), q.numpy(), err_msg=f"{tuple(x0.shape)} {ratio}")
            np.testing.assert_array_equal(y.cpu().numpy(), want.numpy(), err_msg=f"{tuple(x0.shape)} {ratio}")


def test_dpm_step_op_bit_exact(env):
    """One post-network step (prediction conversion, thresholding, first/second-order update) against the same fp32
    operations in torch on the CPU, for every prediction type / order / algorithm; learned-sigma outputs use channel 0."""
    from diffusion_models_dsdiff_amd import _lib
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.sampler import DpmSchedule
    gen = torch.Generator().manual_seed(11)
    B, H, W = 3, 24, 40
    for pred, data_pred, thr, order, Cm in [(0, 1, 1, 1, 1), (0, 1, 1, 2, 2), (1, 1, 0, 2, 1), (2, 1, 1, 2, 1),
                                            (2, 0, 0, 2, 1), (0, 0, 0, 1, 1), (0, 0, 1, 0, 1), (1, 1, 1, 0, 2)]:
        coef = np.zeros((2, _lib.DSD_NCOEF), np.float32)
        coef[:, :6] = np.asarray([0.31, 0.95, 0.87, -0.42, -0.21, 1.37], np.float32)
        sc = DpmSchedule(pred, data_pred, thr, 0.995, 1.0, coef, [10.0, 5.0], [1, order])
        out = torch.randn(B, Cm, H, W, generator=gen)
        x = torch.randn(B, 1, H, W, generator=gen) * 2
        m1 = torch.randn(B, 1, H, W, generator=gen)
        xd, mc, od, m1d = x.cuda().clone(), torch.empty(B, 1, H, W, device="cuda"), out.cuda(), m1.cuda()
        _lib.check(_lib.lib().dsd_op_dpm_step(C.byref(sc.c), 1, _lib.dptr(od), Cm, _lib.dptr(xd), _lib.dptr(mc),
                                              _lib.dptr(m1d), B, H, W, _lib.stream_ptr()))
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        alpha, sigma, cx, cm, cd, ir0 = (f(v) for v in coef[1, :6])
        o = out[:, :1]
        eps = o if pred == 0 else ((x - alpha * o) / sigma if pred == 1 else alpha * o + sigma * x)
        data = bool(data_pred) or order == 0
        m = (x - sigma * eps) / alpha if data else eps
        if thr and data:
            m = ODPM.dynamic_threshold(m)
        want = m if order == 0 else (cx * x - cm * m if order == 1 else (cx * x - cm * m) - cd * (ir0 * (m - m1)))
        tag = f"pred={pred} data={data_pred} thr={thr} order={order} Cm={Cm}"
        np.testing.assert_array_equal(mc.cpu().numpy(), m.numpy(), err_msg=tag)
        np.testing.assert_array_equal(xd.cpu().numpy(), want.numpy(), err_msg=tag)


def test_dpm_solver_all_branches_vs_reference(env):
    """DPM_Solver(...).sample on the device loop for every fixture case (both solver copies of the reference: logSNR /
    uniform / quadratic spacing, orders 1-2, lower-order final, thresholding, denoise-to-zero, dpmsolver & taylor, sub-range)."""
    g, wrap, cond, xT = env
    for key in DPM_CASES:
        sol, kw = _solver(key, wrap, model_kwargs=dict(c_concat=[cond]))
        y = sol.sample(xT, **kw)
        assert rel_l2(y, g[key + "_y"]) < TOL, key
        assert torch.equal(y, sol.sample(xT, **kw)), key                     # deterministic


def test_reference_entry_points(env):
    """GaussianDiffusion.dpm_solver_sample_loop (gaussian_diffusion.py:467-522) and DPMSolverSampler.sample
    (ldm/models/diffusion/dpm_solver_new/sampler.py:35-103) / DDPMModel.log_images(sampler='dpm')."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd.ldm.models.diffusion.dpm_solver_new import DPMSolverSampler
    from diffusion_models_dsdiff_amd.trainers.trainer_ddpm import DDPMModel
    g, wrap, cond, xT = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="eps")
    y = d.dpm_solver_sample_loop(wrap, SHAPE, model_kwargs=dict(c_concat=[cond]), noise=xT)
    assert rel_l2(y, g["A_dpm_20_y"]) < TOL
    d = create_gaussian_diffusion(steps=1000, noise_schedule="cosine", timestep_respacing="12", rescale_timesteps=True)
    y = d.dpm_solver_sample_loop(wrap, SHAPE, model_kwargs=dict(c_concat=[cond]), noise=xT)
    assert rel_l2(y, g["A_dpm_cos_12_y"]) < TOL
    m = DDPMModel(timesteps=1000, parameterization="v").cuda()
    m.model = wrap
    y, _ = DPMSolverSampler(m).sample(10, 2, SHAPE[1:], cond, x_T=xT)
    assert rel_l2(y, g["B_v_uniform_10_y"]) < TOL
    y, _ = DPMSolverSampler(m).sample(6, 2, SHAPE[1:], dict(c_concat=[cond]), x_T=xT)
    assert rel_l2(y, g["B_v_uniform_6_lof_y"]) < TOL
    torch.manual_seed(0)
    log = m.log_images({"t1ce": xT, "image": cond}, N=2, sampler="dpm", ddim_steps=6, pred_mode=True,
                       unconditional_guidance_scale=1.0, ddim_use_original_steps=False, use_ema_scope=False)
    assert log["samples"].shape == SHAPE and bool(torch.isfinite(log["samples"]).all())


def test_generic_callable_path(env):
    """A foreign callable goes through the python loop + fused HIP step per evaluation: same numbers."""
    g, wrap, cond, xT = env
    closure = lambda x, t, **kw: wrap(x, t, c_concat=[cond])
    for key in ("A_dpm_20", "B_eps_dz_6", "B_v_taylor_8"):
        sol, kw = _solver(key, closure)
        assert rel_l2(sol.sample(xT, **kw), g[key + "_y"]) < TOL, key


def test_oracle_agrees_on_unseen_case(env):
    """A configuration with no fixture (3 conditions = C_in 4 is covered elsewhere; here: 2-channel learned-sigma network,
    batch 3, 48x32, 9 logSNR steps with thresholding) against the oracle loop."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion import sampler as dsa
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.model import DSUnetModel
    from oracle import unet as O
    gm = golden("model")
    params = json.loads(str(gm["tinyfilm_cfg"]))
    cfg, sd = O.UNetConfig.from_params(params), fixture_params(gm, "tinyfilm")
    m = DSUnetModel(**params)
    m.load_state_dict(sd, strict=True)
    shape = (3, 1, 48, 32)
    cond3, xT = cond_image((3, 3, 48, 32), 31), randn(shape, 32)
    betas = dpm_case_betas(("A", "linear", "50"))
    ns = dsa.NoiseScheduleVP("discrete", **betas)
    sol = dsa.DPM_Solver(dsa.model_wrapper(m, ns, model_kwargs=dict(c_concat=[cond3.cuda()])), ns,
                         correcting_x0_fn="dynamic_thresholding")
    y = sol.sample(xT.cuda(), steps=9, order=2, skip_type="logSNR", lower_order_final=True)
    net = lambda x, t: O.unet_forward(cfg, sd, torch.cat([x, cond3], 1), t)[0][:, :1]
    want = ODPM.dpm_multistep(net, ODPM.NoiseSchedule(**betas), xT.clone(), steps=9, order=2, skip_type="logSNR",
                              thresholding=True, lower_order_final=True)
    assert rel_l2(y, want) < TOL
