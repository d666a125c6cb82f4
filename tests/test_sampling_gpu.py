"""Sampling-loop parity (GPU): dsd_sample through the reference call signatures vs fixtures produced by the
reference's own loops (tests/golden/gen_golden.py) and vs the oracle.  Tolerance: rel-L2 <= 1e-4 on the final fp32 image
(BASELINE.json north_star); measured values are ~1e-6."""
import json

import pytest
import torch

from oracle import samplers as OS, unet as O
from util import golden, fixture_params, rel_l2, randn, cond_image

pytestmark = pytest.mark.gpu
TOL = 1e-4
SHAPE = (2, 1, 32, 32)


@pytest.fixture(scope="module")
def env():
    from diffusion_models_dsdiff_amd import _lib
    from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddpm import DiffusionWrapper
    _lib.require_gpu(0)
    gm, gl = golden("model"), golden("loops")
    params = json.loads(str(gm["tiny_cfg"]))
    wrap = DiffusionWrapper({"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}, "concat")
    wrap.diffusion_model.load_state_dict(fixture_params(gm, "tiny"), strict=True)
    cond = cond_image(SHAPE, int(gl["cond_seed"])).cuda()
    xT = randn(SHAPE, int(gl["xT_seed"])).cuda()
    return gl, wrap, cond, xT, params


def noise_for(gl, key, steps):
    return randn((steps,) + SHAPE, int(gl[key + "_noise_seed"])).cuda()


def test_family_a_ddpm_and_ddim(env):
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="50", rescale_timesteps=True, parameterization="v")
    kw = dict(noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond]))
    y = d.p_sample_loop(wrap, SHAPE, step_noise=noise_for(gl, "A_ddpm_50", 50), **kw)
    assert rel_l2(y, gl["A_ddpm_50_y"]) < TOL
    y = d.ddim_sample_loop(wrap, SHAPE, eta=0.0, step_noise=noise_for(gl, "A_ddim_50", 50), **kw)
    assert rel_l2(y, gl["A_ddim_50_y"]) < TOL
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v")
    y = d.ddim_sample_loop(wrap, SHAPE, eta=1.0, step_noise=noise_for(gl, "A_ddim_20_eta1", 20), **kw)
    assert rel_l2(y, gl["A_ddim_20_eta1_y"]) < TOL


def test_family_a_1000_step_ddpm_headline(env):
    """BASELINE config shape of the loop (1000-step DDPM, v-param, clip) on the tiny model."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="", rescale_timesteps=False, parameterization="v")
    y = d.p_sample_loop(wrap, SHAPE, noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond]),
                        step_noise=noise_for(gl, "A_ddpm_1000", 1000))
    assert rel_l2(y, gl["A_ddpm_1000_y"]) < TOL


def test_family_b_ddpm_and_ddim(env):
    from diffusion_models_dsdiff_amd.trainers.trainer_ddpm import DDPMModel
    from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddim import DDIMSampler
    gl, wrap, cond, xT, params = env
    m = DDPMModel(timesteps=50, parameterization="v", clip_denoised=True).cuda()
    m.model = wrap
    y = m.p_sample_loop(SHAPE, dict(c_concat=[cond]), x_T=xT, step_noise=noise_for(gl, "B_ddpm_50", 50))
    assert rel_l2(y, gl["B_ddpm_50_y"]) < TOL
    m = DDPMModel(timesteps=1000, parameterization="v").cuda()
    m.model = wrap
    s = DDIMSampler(m)
    y, _ = s.sample(20, 2, SHAPE[1:], dict(c_concat=[cond]), eta=0.0, verbose=False, x_T=xT,
                    step_noise=noise_for(gl, "B_ddim_20", 20))
    assert rel_l2(y, gl["B_ddim_20_y"]) < TOL
    y, _ = s.sample(20, 2, SHAPE[1:], dict(c_concat=[cond]), eta=1.0, verbose=False, x_T=xT,
                    step_noise=noise_for(gl, "B_ddim_20_eta1", 20))
    assert rel_l2(y, gl["B_ddim_20_eta1_y"]) < TOL


def test_learned_range_film_model_four_channels(env):
    """Next-row f-1: LEARNED_RANGE variance + FiLM ResBlocks + resblock_updown, 3 conditions (BraTS-like C_in = 4)."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.model import DSUnetModel
    gl, _, _, xT, _ = env
    gm = golden("model")
    m = DSUnetModel(**json.loads(str(gm["tinyfilm_cfg"])))
    m.load_state_dict(fixture_params(gm, "tinyfilm"), strict=True)
    cond3 = cond_image((2, 3, 32, 32), int(gl["cond3_seed"])).cuda()
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing="20", rescale_timesteps=True)
    kw = dict(noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond3]))
    y = d.p_sample_loop(m, SHAPE, step_noise=noise_for(gl, "A_lr_ddpm_20", 20), **kw)
    assert rel_l2(y, gl["A_lr_ddpm_20_y"]) < TOL
    y = d.ddim_sample_loop(m, SHAPE, eta=0.0, step_noise=noise_for(gl, "A_lr_ddim_20", 20), **kw)
    assert rel_l2(y, gl["A_lr_ddim_20_y"]) < TOL


def test_denoised_fn_and_cond_fn_hooks(env):
    """gaussian_diffusion.py:312-313 (denoised_fn on the predicted x_start, before the clip), :386-398 / :460-463 (cond_fn shifts the
    DDPM mean by variance * gradient; SpacedDiffusion hands it the wrapped model timestep) — against loops the REFERENCE ran
    with the same two functions (tests/golden/hooks.npz).  The hooks run in Python between the network and the fused update
    kernel; cond_fn + DDIM and cond_fn + learned variance raise and name the loop to use."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    _, wrap, cond, xT, _ = env
    gh = golden("hooks")
    den = lambda x: torch.tanh(1.5 * x)
    cfn = lambda x, t, **kw: -0.3 * (x - kw["c_concat"][0]) * (1.0 + t.float().view(-1, 1, 1, 1) / 1000.0)
    cases = [("ddpm_20_v_denoised", dict(timestep_respacing="20", parameterization="v"), "p_sample_loop", dict(denoised_fn=den)),
             ("ddim_20_v_denoised_eta05", dict(timestep_respacing="20", parameterization="v"), "ddim_sample_loop", dict(denoised_fn=den, eta=0.5)),
             ("ddpm_20_eps_small_denoised", dict(timestep_respacing="20", sigma_small=True), "p_sample_loop", dict(denoised_fn=den)),
             ("ddpm_20_v_cond", dict(timestep_respacing="20", parameterization="v"), "p_sample_loop", dict(cond_fn=cfn)),
             ("ddpm_25_eps_small_cond_denoised", dict(timestep_respacing="25", sigma_small=True), "p_sample_loop",
              dict(cond_fn=cfn, denoised_fn=den))]
    for key, dkw, fn, kw in cases:
        d = create_gaussian_diffusion(steps=1000, rescale_timesteps=True, **dkw)
        z = randn((d.num_timesteps,) + SHAPE, int(gh[key + "_noise_seed"])).cuda()
        for model in (wrap, wrap.diffusion_model):       # the DiffusionWrapper and the bare native U-Net
            y = getattr(d, fn)(model, SHAPE, noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond]), step_noise=z, **kw)
            assert rel_l2(y, gh[key + "_y"]) < TOL, key
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v")
    with pytest.raises(NotImplementedError, match="p_sample_loop"):
        d.ddim_sample_loop(wrap, SHAPE, noise=xT, model_kwargs=dict(c_concat=[cond]), cond_fn=cfn)
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, learn_sigma=True)
    with pytest.raises(NotImplementedError, match="learn_sigma"):
        d.p_sample_loop(wrap, SHAPE, noise=xT, model_kwargs=dict(c_concat=[cond]), cond_fn=cfn)


def test_generic_callable_and_single_step_paths(env):
    """A foreign callable goes through the python loop + fused HIP update; p_sample returns pred_xstart."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="50", rescale_timesteps=True, parameterization="v")
    closure = lambda x, t, **kw: wrap(x, t, c_concat=[cond])[0]
    y = d.p_sample_loop(closure, SHAPE, noise=xT, step_noise=noise_for(gl, "A_ddpm_50", 50), device="cuda")
    assert rel_l2(y, gl["A_ddpm_50_y"]) < TOL
    out = d.p_sample(closure, xT, torch.tensor([49, 49]).cuda())
    assert out["sample"].shape == SHAPE and float(out["pred_xstart"].abs().max()) <= 1.0
    last = None
    for last in d.ddim_sample_loop_progressive(closure, SHAPE, noise=xT, device="cuda"):
        pass
    assert last["sample"].shape == SHAPE


def test_philox_sampling_is_seeded_and_batch_independent(env):
    """Device Philox noise: reproducible under torch.manual_seed; every slice is an independent chain."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v")
    kw = dict(noise=xT, model_kwargs=dict(c_concat=[cond]))
    a = d.p_sample_loop(wrap, SHAPE, seed=123, **kw)
    b = d.p_sample_loop(wrap, SHAPE, seed=123, **kw)
    c = d.p_sample_loop(wrap, SHAPE, seed=124, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    torch.manual_seed(7)
    e = d.p_sample_loop(wrap, SHAPE, **kw)
    torch.manual_seed(7)
    f = d.p_sample_loop(wrap, SHAPE, **kw)
    assert torch.equal(e, f)
    # DDIM eta=0 is deterministic: running sample 0 alone equals its row in the batch (no cross-sample op)
    full = d.ddim_sample_loop(wrap, SHAPE, **kw)
    one = d.ddim_sample_loop(wrap, (1,) + SHAPE[1:], noise=xT[:1], model_kwargs=dict(c_concat=[cond[:1]]))
    assert rel_l2(one, full[:1]) < 1e-5
    assert bool(torch.isfinite(a).all())


def test_share_zero_streams_is_bit_identical(env):
    """Optional dedup of the two all-zero-input streams (dsd_set_share_zero_streams): the same result with fewer FLOPs."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v")
    unet = wrap.diffusion_model
    shape = (4, 1, 32, 32)
    c4, x4 = torch.cat([cond, cond.flip(0)]), torch.cat([xT, xT.flip(0)])
    kw = dict(noise=x4, model_kwargs=dict(c_concat=[c4]), seed=77)
    for prec in ("bf16x6", "f32"):
        unet.set_precision(prec)
        unet.share_zero_streams(False)
        a = d.p_sample_loop(wrap, shape, **kw)
        f0 = unet.plan_info()["flops"]
        unet.share_zero_streams(True)
        b = d.p_sample_loop(wrap, shape, **kw)
        f1 = unet.plan_info()["flops"]
        unet.share_zero_streams(False)
        # f32: no kernel's arithmetic depends on the batch -> same bits.  Split modes: the GroupNorm statistics ride in the
        # convolution epilogues (fp32 partials per 32-row block, fp64 above), and the batch-1 streams pick other tiles than
        # the batch-4 ones, so the sums are ordered differently: equal to fp64-rounding of the statistics only
        if prec == "f32":
            assert torch.equal(a, b), prec
        else:
            assert rel_l2(b, a) < 1e-6, prec
        assert f1 < 0.85 * f0
    unet.set_precision("bf16x6")


def test_more_sampler_branches(env):
    """eps / x0 prediction, FIXED_SMALL variance, clip off, DDIM eta 0.5, 'ddimN' striding (family A); eps-parameterised and
    ddim_use_original_steps DDIM (family B) — fixtures from the reference's own loops (tests/golden/loops2.npz)."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd.trainers.trainer_ddpm import DDPMModel
    from diffusion_models_dsdiff_amd.ldm.models.diffusion.ddim import DDIMSampler
    gl, wrap, cond, xT, _ = env
    gl2 = golden("loops2")
    mk = create_gaussian_diffusion
    cases = [("A_eps_ddpm_20", mk(steps=1000, timestep_respacing="20", rescale_timesteps=True), "p_sample_loop", {}),
             ("A_eps_small_ddpm_20", mk(steps=1000, timestep_respacing="20", rescale_timesteps=True, sigma_small=True), "p_sample_loop", {}),
             ("A_x0_ddpm_20", mk(steps=1000, timestep_respacing="20", rescale_timesteps=True, predict_xstart=True), "p_sample_loop", {}),
             ("A_eps_ddim_20_eta05", mk(steps=1000, timestep_respacing="20", rescale_timesteps=True), "ddim_sample_loop", {"eta": 0.5}),
             ("A_x0_ddim_ddim25", mk(steps=1000, timestep_respacing="ddim25", predict_xstart=True), "ddim_sample_loop", {"eta": 0.0}),
             ("A_v_noclip_ddpm_20", mk(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v"),
              "p_sample_loop", {"clip_denoised": False})]
    for key, d, fn, kw in cases:
        z = randn((d.num_timesteps,) + SHAPE, int(gl2[key + "_noise_seed"])).cuda()
        y = getattr(d, fn)(wrap, SHAPE, noise=xT, model_kwargs=dict(c_concat=[cond]), step_noise=z, **kw)
        assert rel_l2(y, gl2[key + "_y"]) < TOL, key
    for key, param, S_, orig, eta in (("B_eps_ddim_10", "eps", 10, False, 0.0), ("B_v_ddim_orig50_eta1", "v", 10, True, 1.0),
                                      ("B_eps_ddim_orig50", "eps", 10, True, 0.0)):
        m = DDPMModel(timesteps=50, parameterization=param).cuda()
        m.model = wrap
        n = 50 if orig else S_
        z = randn((n,) + SHAPE, int(gl2[key + "_noise_seed"])).cuda()
        y, _ = DDIMSampler(m).sample(S_, 2, SHAPE[1:], dict(c_concat=[cond]), eta=eta, verbose=False, x_T=xT, step_noise=z,
                                     ddim_use_original_steps=orig)
        assert rel_l2(y, gl2[key + "_y"]) < TOL, key


def test_baseline_config_0_and_2_shapes(env):
    """BASELINE.json configs[0] (64x64 1-ch slice, 50-step DDPM through trainers/trainer_ddpm.py's sampler = family B) and
    configs[2] (BraTS: 3 conditioning modalities -> T1ce, 50-step DDIM, learned sigma + FiLM network of
    v2-1-cddpm-ds-disc-openai-diffusion.yaml) on the tiny networks against the oracle loops.  BraTS slices are 240x240 in the
    raw data; the reference itself cannot run them (five stride-2 stages need multiples of 32) and trains
    on 256-padded slices, so 240 must fail loudly and 256-compatible sizes must match."""
    from diffusion_models_dsdiff_amd.trainers.trainer_ddpm import DDPMModel
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.model import DSUnetModel
    from diffusion_models_dsdiff_amd import _lib
    gl, wrap, _, _, params = env
    gm = golden("model")
    # configs[0]
    shape = (1, 1, 64, 64)
    cond, xT = cond_image(shape, 71), randn(shape, 72)
    z = randn((50,) + shape, 73)
    m = DDPMModel(timesteps=50, parameterization="v", clip_denoised=True).cuda()
    m.model = wrap
    y = m.p_sample_loop(shape, dict(c_concat=[cond.cuda()]), x_T=xT.cuda(), step_noise=z.cuda())
    cfg, sd = O.UNetConfig.from_params(params), fixture_params(gm, "tiny")
    net = lambda x, t: O.unet_forward(cfg, sd, x, t)[0]
    want = OS.DiffusionB(timesteps=50, parameterization="v").p_sample_loop(net, xT, z, [cond])
    assert rel_l2(y, want) < TOL
    # configs[2]
    fparams = json.loads(str(gm["tinyfilm_cfg"]))
    fm = DSUnetModel(**fparams)
    fsd = fixture_params(gm, "tinyfilm")
    fm.load_state_dict(fsd, strict=True)
    shape = (2, 1, 64, 32)
    cond3, xT = cond_image((2, 3, 64, 32), 74), randn(shape, 75)
    z = randn((50,) + shape, 76)
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing="50", rescale_timesteps=True)
    y = d.ddim_sample_loop(fm, shape, noise=xT.cuda(), clip_denoised=True, model_kwargs=dict(c_concat=[cond3.cuda()]), eta=0.0,
                           step_noise=z.cuda())
    fcfg = O.UNetConfig.from_params(fparams)
    fnet = lambda x, t: O.unet_forward(fcfg, fsd, x, t)[0]
    od = OS.DiffusionA(steps=1000, timestep_respacing="50", rescale_timesteps=True, learn_sigma=True)
    assert rel_l2(y, od.ddim_sample_loop(fnet, xT, z, [cond3], eta=0.0)) < TOL
    # the tiny network has two stride-2 stages (multiples of 4); 30x30 is to it what 240x240 is to the 6-level network
    with pytest.raises((_lib.DsdError, AssertionError, RuntimeError)):
        d.ddim_sample_loop(fm, (1, 1, 30, 30), noise=torch.zeros(1, 1, 30, 30).cuda(), clip_denoised=True,
                           model_kwargs=dict(c_concat=[torch.zeros(1, 3, 30, 30).cuda()]), eta=0.0)


def test_philox_noise_is_keyed_by_slice_not_by_batch_position(env):
    """with dsd_set_slice_ids the on-device noise of a slice depends on (seed, step, slice index) only, so a
    volume sampled as one batch, as two shards ([r::2], what 2 ranks do) or slice by slice gives the same images."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, _, _, _ = env
    unet = wrap.diffusion_model
    n = 5
    shape = (n, 1, 32, 32)
    cond, xT = cond_image(shape, 700).cuda(), randn(shape, 701).cuda()
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="10", rescale_timesteps=True, parameterization="v")

    def run(idx):
        idx = list(idx)
        unet.set_slice_ids(idx)
        y = d.p_sample_loop(wrap, (len(idx), 1, 32, 32), noise=xT[idx], clip_denoised=True,
                            model_kwargs=dict(c_concat=[cond[idx]]), seed=1234)
        unet.set_slice_ids(None)
        return y
    whole = run(range(n))
    for r in range(2):                                       # the [r::R] shards of parallel.shard_indices
        idx = list(range(r, n, 2))
        part = run(idx)
        assert rel_l2(part, whole[idx]) < 1e-6               # kernels may tile a smaller batch differently: fp32 rounding
    single = run([3])
    assert rel_l2(single, whole[3:4]) < 1e-6
    # without ids the noise follows the batch position (the old behaviour): slice 3 alone then differs from slice 3 in the batch
    y = d.p_sample_loop(wrap, (1, 1, 32, 32), noise=xT[3:4], clip_denoised=True, model_kwargs=dict(c_concat=[cond[3:4]]), seed=1234)
    assert rel_l2(y, whole[3:4]) > 1e-3
    with pytest.raises(Exception):                           # ids must match the batch
        unet.set_slice_ids([0, 1])
        d.p_sample_loop(wrap, (n, 1, 32, 32), noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond]), seed=1)
    unet.set_slice_ids(None)


def test_graph_replay_matches_host_launches(env):
    """hipGraph replay of the network evaluation inside dsd_sample / dsd_sample_dpm is bit-identical to host launches."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gl, wrap, cond, xT, _ = env
    unet = wrap.diffusion_model
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True, parameterization="v")
    kw = dict(noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond]))
    out = {}
    for on in (False, True):
        unet.use_graph(on)
        s0 = unet.graph_stats()
        out[on] = (d.p_sample_loop(wrap, SHAPE, seed=5, **kw), d.ddim_sample_loop(wrap, SHAPE, eta=0.5, seed=6, **kw),
                   d.dpm_solver_sample_loop(wrap, SHAPE, model_kwargs=dict(c_concat=[cond]), noise=xT))
        s1 = unet.graph_stats()
        assert (s1["launches"] > s0["launches"]) == on
    unet.use_graph(False)
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
