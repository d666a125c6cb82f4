"""Host-side product logic (no GPU): the shim's integer timestep maps / float64 tables / schedule packing
against the reference-generated fixtures — bit-exact for the integer work."""
import hashlib
import json

import numpy as np
import pytest

from diffusion_models_dsdiff_amd import _lib
from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion import gaussian_diffusion as gd
from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.respace import SpacedDiffusion, space_timesteps
from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddim import DDIMSampler, make_ddim_timesteps
from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddpm import DDPM, make_beta_schedule
from diffusion_models_dsdiff_amd.trainers.trainer_ddpm import DDPMModel
from oracle import schedules as S
from util import golden


def mk(T, spec, **kw):
    return create_gaussian_diffusion(steps=T, timestep_respacing=spec if isinstance(spec, str) else "", **kw)


@pytest.mark.parametrize("T,spec,key", [
    (1000, "20", "A_1000_20"), (1000, "50", "A_1000_50"), (1000, "ddim50", "A_1000_ddim50"),
    (1000, "100", "A_1000_100"), (1000, [1000], "A_1000_full"), (2000, "100", "A_2000_100"),
    (1000, "10,20,30", "A_1000_10_20_30"), (1000, "250", "A_1000_250"), (50, "50", "A_50_50")])
def test_timestep_map_bit_exact(T, spec, key):
    g = golden("schedules")
    d = mk(T, spec)
    tm = np.asarray(d.timestep_map, dtype=np.int64)
    assert np.array_equal(tm, g[key + "_map"])
    assert hashlib.sha256(tm.tobytes()).hexdigest()[:16] == json.loads(str(g["hashes_json"]))[key]
    # what the device loop hands to the network: exact integers in fp32, descending
    s = d._schedule(False, 0.0, True)
    assert np.array_equal(s.t_model.astype(np.int64), tm[::-1])
    assert s.nonzero.tolist() == [1] * (len(tm) - 1) + [0]


@pytest.mark.parametrize("T,spec,key", [(1000, "20", "A_1000_20"), (1000, "50", "A_1000_50"),
                                        (1000, [1000], "A_1000_full"), (2000, "100", "A_2000_100")])
def test_float64_tables_equal_reference(T, spec, key):
    g = golden("schedules")
    d = mk(T, spec)
    for nm in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
               "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
               "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(getattr(d, nm), g[f"{key}_{nm}"]), nm


def test_rescaled_timesteps_are_fp32_products():
    d = mk(1000, "50", rescale_timesteps=True)
    assert np.array_equal(d._model_timestep_values(), np.asarray(d.timestep_map, np.float32))   # 1000/1000 = 1
    d = mk(2000, "100", rescale_timesteps=True)
    assert np.array_equal(d._model_timestep_values(), np.asarray(d.timestep_map, np.float32) * np.float32(0.5))


def test_schedule_rows_match_oracle_tables():
    d = mk(1000, "50", parameterization="v")
    s = d._schedule(False, 0.0, True)
    betas, _ = S.spaced(S.named_beta_schedule("linear", 1000), S.space_timesteps(1000, "50"))
    tab = S.gaussian_tables(betas)
    idx = np.arange(49, -1, -1)
    assert np.array_equal(s.coef[:, 0], tab["sqrt_alphas_cumprod"][idx].astype(np.float32))
    assert np.array_equal(s.coef[:, 4], tab["posterior_mean_coef1"][idx].astype(np.float32))
    lv = np.log(np.append(tab["posterior_variance"][1], betas[1:]))
    assert np.array_equal(s.coef[:, 6], lv[idx].astype(np.float32))
    assert s.c.mode == _lib.MODE_A_DDPM and s.c.pred == _lib.PRED_V and s.c.learned_range == 0
    s = d._schedule(True, 0.5, False)
    assert s.c.mode == _lib.MODE_A_DDIM and abs(s.c.eta - 0.5) < 1e-7 and s.c.clip_denoised == 0
    assert np.array_equal(s.coef[:, 5], tab["alphas_cumprod_prev"][idx].astype(np.float32))
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing="20")
    s = d._schedule(False, 0.0, True)
    assert s.c.learned_range == 1 and s.c.pred == _lib.PRED_EPS
    assert np.array_equal(s.coef[:, 7], np.log(d.betas)[::-1].astype(np.float32))


def test_family_b_tables_and_ddim():
    g = golden("schedules")
    for T in (50, 1000, 2000):
        assert np.array_equal(make_beta_schedule("linear", T), g[f"B_linear_{T}_betas"])
    assert np.array_equal(make_beta_schedule("sqrt_linear", 1000), g["B_sqrt_linear_1000_betas"])
    for n, T in [(50, 1000), (20, 2000), (20, 1000), (7, 50)]:
        assert np.array_equal(make_ddim_timesteps("uniform", n, T, verbose=False), g[f"B_ddim_uniform_{n}_{T}"])
    assert np.array_equal(make_ddim_timesteps("quad", 20, 1000, verbose=False), g["B_ddim_quad_20_1000"])
    m = DDPMModel(timesteps=1000, parameterization="v")
    tab = S.ldm_tables(S.make_beta_schedule("linear", 1000))
    for k, v in tab.items():
        assert np.array_equal(getattr(m, k).numpy(), v), k
    sm = DDIMSampler(m)
    for eta in (0.0, 1.0):
        sm.make_schedule(20, ddim_eta=eta, verbose=False)
        assert np.array_equal(np.asarray(sm.ddim_alphas, np.float64), g[f"B_ddim_params_eta{int(eta)}_alphas"])
        assert np.array_equal(np.asarray(sm.ddim_alphas_prev, np.float64), g[f"B_ddim_params_eta{int(eta)}_alphas_prev"])
        np.testing.assert_allclose(sm.ddim_sigmas, g[f"B_ddim_params_eta{int(eta)}_sigmas"], rtol=2e-6)
        sc = sm._schedule(False, True)
        assert sc.c.mode == _lib.MODE_B_DDIM and sc.steps == 20
        assert np.array_equal(sc.t_model.astype(np.int64), g["B_ddim_uniform_20_1000"][::-1])
    s = m._schedule(True)
    assert s.c.mode == _lib.MODE_B_DDPM and s.steps == 1000 and s.t_model[0] == 999 and s.nonzero[-1] == 0


def test_instantiate_from_config_resolves_reference_targets():
    from diffusion_models_dsdiff_amd.ldm.util import get_obj_from_str
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.model import DSUnetModel
    assert get_obj_from_str("UNet_DS_Diff.model.DSUnetModel") is DSUnetModel
    assert get_obj_from_str("ldm.models.diffusion.ddim.DDIMSampler") is DDIMSampler


def test_errors_mirror_reference():
    with pytest.raises(ValueError):
        space_timesteps(10, "20")
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(NotImplementedError):
        gd.get_named_beta_schedule("nope", 10)
    with pytest.raises(AssertionError):
        gd.GaussianDiffusion(betas=np.array([0.0, 0.1]), model_mean_type=gd.ModelMeanType.EPSILON,
                             model_var_type=gd.ModelVarType.FIXED_LARGE, loss_type=gd.LossType.MSE)


# ---------------------------------------------------------------------------------------- DPM-Solver(++) host side
def _dpm_solver_for(key):
    import torch
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion import sampler as dsa
    from util import DPM_CASES, dpm_case_betas
    src, mtype, kw = DPM_CASES[key]
    kw = dict(kw)
    ns = dsa.NoiseScheduleVP("discrete", **dpm_case_betas(src))
    fn = dsa.model_wrapper(lambda x, t, **k: x, ns, model_type=mtype)
    sol = dsa.DPM_Solver(fn, ns, algorithm_type=kw.pop("algorithm", "dpmsolver++"),
                         correcting_x0_fn="dynamic_thresholding" if kw.pop("thresholding", False) else None)
    return dsa, ns, sol, mtype, kw


def test_dpm_noise_schedule_bit_exact():
    """Product NoiseScheduleVP / get_time_steps against the reference's fp32 values (tests/golden/dpm.npz)."""
    from util import DPM_CASES
    g = golden("dpm")
    for key in DPM_CASES:
        _, ns, sol, _, kw = _dpm_solver_for(key)
        assert ns.total_N == int(g[key + "_totalN"])
        ts = sol.get_time_steps(kw["skip_type"], kw.get("t_start") or ns.T, kw.get("t_end") or 1. / ns.total_N, kw["steps"])
        np.testing.assert_array_equal(ts.numpy(), g[key + "_ts"], err_msg=key)
        np.testing.assert_array_equal(ns.marginal_alpha(ts).numpy(), g[key + "_alpha"], err_msg=key)
        np.testing.assert_array_equal(ns.marginal_std(ts).numpy(), g[key + "_std"], err_msg=key)
        np.testing.assert_array_equal(ns.marginal_lambda(ts).numpy(), g[key + "_lam"], err_msg=key)
    import torch
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.sampler import NoiseScheduleVP
    ns = NoiseScheduleVP("discrete", betas=torch.from_numpy(S.named_beta_schedule("cosine", 1000)).float())
    assert ns.total_N == int(g["A_cos1000_totalN"]) < 1000             # numerical_clip_alpha


def test_dpm_schedule_rows_reproduce_the_oracle_loop():
    """The coefficient rows handed to dsd_sample_dpm, replayed on the CPU with the device kernels' fp32 operation order
    (emulated below with torch fp32 ops), must reproduce the oracle's multistep loop on an analytic network: this pins
    order selection, lower-order-final, denoise-to-zero and every coefficient without a GPU."""
    import torch
    from oracle import dpm as odpm
    from util import DPM_CASES, dpm_case_betas
    net = lambda x, t: 0.3 * x + torch.sin(t / 1000.)[:, None, None, None] + 0.05 * x.flip(-1)
    x_T = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 1, 8, 8)).astype(np.float32))
    for key, (src, mtype, okw) in DPM_CASES.items():
        _, ns, sol, _, kw = _dpm_solver_for(key)
        sc = sol.build_schedule(kw.pop("steps"), kw.pop("t_start", None), kw.pop("t_end", None), **kw)
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        x, m_prev = x_T.clone(), None
        for k in range(sc.steps):
            alpha, sigma, cx, cm, cd, ir0 = (f(v) for v in sc.coef[k, :6])
            out = net(x, torch.full((2,), float(sc.t_input[k])))
            eps = out if mtype == "noise" else ((x - alpha * out) / sigma if mtype == "x_start" else alpha * out + sigma * x)
            data = bool(sc.c.data_pred) or sc.order[k] == 0
            m = (x - sigma * eps) / alpha if data else eps
            if sc.c.thresholding and data:
                m = odpm.dynamic_threshold(m, sc.c.threshold_ratio, sc.c.threshold_max)
            if sc.order[k] == 0:
                x = m
            else:
                x_new = cx * x - cm * m
                if sc.order[k] == 2:
                    x_new = x_new - cd * (ir0 * (m - m_prev))
                x = x_new
            m_prev = m
        want = odpm.dpm_multistep(net, odpm.NoiseSchedule(**dpm_case_betas(src)), x_T.clone(), model_type=mtype, **okw)
        np.testing.assert_array_equal(x.numpy(), want.numpy(), err_msg=key)


def test_dpm_unsupported_options_raise():
    dsa, ns, sol, _, _ = _dpm_solver_for("B_eps_uniform_6")
    import torch
    x = torch.zeros(1, 1, 8, 8)
    with pytest.raises(NotImplementedError):
        sol.sample(x, method="singlestep")
    with pytest.raises(NotImplementedError):
        sol.sample(x, order=3)
    with pytest.raises(NotImplementedError):
        dsa.model_wrapper(lambda x, t: x, ns, guidance_type="classifier")
    with pytest.raises(NotImplementedError):
        dsa.DPM_Solver(dsa.model_wrapper(lambda x, t: x, ns), ns, correcting_x0_fn=lambda x0, t: x0)
    with pytest.raises(TypeError):
        dsa.DPM_Solver(lambda x, t: x, ns)
    with pytest.raises(RuntimeError):                                    # CPU tensors never fall back to a CPU loop
        sol.sample(x, steps=6)
    with pytest.raises(ValueError):
        dsa.NoiseScheduleVP("cosine")


# ---------------------------------------------------------------------------------------- convolution planner (host)
def _conv_plan(N, H, W, Cin, Cout, ks=3, stride=1, precision=2):
    import ctypes as C
    st, nt, ksp, sb = C.c_int(), C.c_int(), C.c_int(), C.c_uint64()
    _lib.check(_lib.lib().dsd_conv_plan(N, H, W, Cin, Cout, ks, stride, precision, C.byref(st), C.byref(nt), C.byref(ksp), C.byref(sb)))
    return st.value, nt.value, ksp.value, sb.value


def test_conv_plan_invariants():
    """dsd_conv_plan is pure host logic: structure / tile width / split-K of the layers of the BASELINE network.  The
    headline layers must stay on the 256-row A-direct kernel with the widest tile and no split; small grids split their
    k-loop, ask for exactly ksplit*M*Cout*4 scratch bytes and never more workgroups than the model allows; fp32 mode and
    shapes the split kernels cannot take fall back."""
    for shape in ((16, 256, 256, 320, 320), (16, 256, 256, 640, 320), (16, 128, 128, 320, 320), (16, 64, 64, 640, 640),
                  (16, 32, 32, 640, 640)):
        st, nt, ks, sb = _conv_plan(*shape)
        assert (st, nt, ks, sb) == (2, 5, 1, 0), shape
    for shape in ((16, 8, 8, 960, 960), (16, 8, 8, 1920, 960), (1, 8, 8, 960, 960), (1, 16, 16, 960, 960), (1, 32, 32, 640, 640)):
        N, H, W, Cin, Cout = shape
        st, nt, ks, sb = _conv_plan(*shape)
        assert st in (1, 2) and 1 <= nt <= 5 and 2 <= ks <= 16, (shape, st, nt, ks)
        assert sb == ks * N * H * W * Cout * 4, shape
        assert (9 * Cin // 32) // ks >= 8, shape                       # every chunk keeps at least 8 k-tiles
        rows = 128 * st
        blocks = -(-N * H * W // rows) * -(-(Cout // 32) // nt) * ks
        assert blocks <= 768, (shape, blocks)
    assert _conv_plan(16, 8, 8, 960, 480, ks=1)[2] >= 1                 # K = 30 tiles: at most x3
    assert _conv_plan(16, 256, 256, 320, 320, precision=0)[0] == -1     # fp32 mode: the fp32 kernel
    assert _conv_plan(16, 256, 256, 1, 320)[0] == -1                    # C_in = 1: direct kernel, never split precision
    assert _conv_plan(1, 16, 16, 320, 1)[2] == 1                        # C_out = 1 cannot use the float4 reduction
    assert _conv_plan(16, 256, 256, 320, 320, precision=1)[0] == 2      # bf16x3: 256-row tile on the big layers
    assert _conv_plan(16, 16, 16, 960, 960, precision=1)[0] == 0        # ... staged structure at M = 4096
    assert _conv_plan(16, 256, 256, 320, 320, precision=2 | 16)[0] == 1 # forced structures are honoured
    assert _conv_plan(16, 256, 256, 320, 320, precision=2 | 32)[0] == 0


def test_wrapped_model_hands_the_original_timesteps_to_the_network():
    """_wrap_model (the role of respace.py:116-128): loop index -> timestep of the ORIGINAL process, int64 when the
    timesteps are not rescaled, fp32 (x 1000 / T_orig) otherwise — the same values the device loop uploads."""
    import torch
    seen = {}

    def net(x, t, **kw):
        seen["t"], seen["kw"] = t, kw
        return x

    for rescale in (False, True):
        d = create_gaussian_diffusion(steps=1000, timestep_respacing="50", rescale_timesteps=rescale, parameterization="v")
        tmap = np.asarray(sorted(space_timesteps(1000, "50")), dtype=np.int64)
        w = d._wrap_model(net)
        assert d._wrap_model(w) is w
        idx = torch.tensor([0, 7, 49])
        w(torch.zeros(3, 1), idx, c_concat=[1])
        assert seen["kw"] == {"c_concat": [1]}
        if rescale:
            assert seen["t"].dtype == torch.float32
            want = tmap[idx.numpy()].astype(np.float32) * np.float32(1000.0 / 1000)
            assert np.array_equal(seen["t"].numpy(), want)
        else:
            assert seen["t"].dtype == torch.int64 and np.array_equal(seen["t"].numpy(), tmap[idx.numpy()])
        assert np.array_equal(d._model_timestep_values()[idx.numpy()], seen["t"].numpy().astype(np.float32))


def test_bench_kernel_symbols_match_the_committed_profiles():
    """bench.py maps a plan kind to the kernel symbol rocprofv3 reports (roofline.traffic comes from the PMC summary of that
    symbol): every convolution kind of the committed bench line must be found in the committed summaries, and the bench
    line's dominant kernel must have carried its traffic figure."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    line = json.load(open(os.path.join(root, "profiles", "r02_bench_1gpu.json")))
    pmc = json.load(open(os.path.join(root, "profiles", "r02_pmc.json")))
    stats = json.load(open(os.path.join(root, "profiles", "r02_kernel_stats.json")))
    names = [k.replace(" ", "") for k in pmc if k != "_meta"]
    traced = [e["kernel"].replace(" ", "") for e in stats]
    kinds = [k for k in line["kernels"] if k.startswith("conv_bf16x6")]
    assert len(kinds) >= 5
    for k in kinds:
        sym = bench.kernel_symbol(k)
        assert sym and any(sym in n for n in names), (k, sym)
        assert any(sym in n for n in traced), (k, sym)
    r = line["roofline"]
    assert r["traffic"] and r["traffic_source"]["file"] == "profiles/r02_pmc.json"
    assert bench.kernel_symbol(r["kernel"]) in r["traffic_source"]["kernel"].replace(" ", "")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
