"""Latent path end to end (BASELINE config 4): the plain UNetModel that denoises the VAE latents
(ldm/modules/diffusionmodules/openaimodel.py:571-958) behind the C ABI (DSD_BLOCK_UNET), and the pipeline
encode -> sample in latent space -> decode of trainers/trainer_latent_diffusion.py:183 on native kernels only.
Fixtures: the reference's own UNetModel (tests/golden/latent_unet.npz)."""
import json

import pytest
import torch

from oracle import samplers as OS, unet as O, vae as V
from util import golden, fixture_params, rel_l2, randn

pytestmark = pytest.mark.gpu


def native_unet(key):
    from diffusion_models_dsdiff_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    g = golden("latent_unet")
    params = json.loads(str(g[key + "_cfg"]))
    m = UNetModel(**params)
    ref_names = [n for n, _ in json.loads(str(g[key + "_params"]))]
    assert sorted(m.state_dict().keys()) == sorted(ref_names)          # reference checkpoints load by name
    sd = fixture_params(g, key)
    m.load_state_dict(sd, strict=True)
    return g, m, params, sd


@pytest.mark.parametrize("key", ["lu", "lu2"])
def test_unet_model_golden(key):
    g, m, params, sd = native_unet(key)
    x = randn(tuple(int(v) for v in g[key + "_xshape"]), int(g[key + "_seed"]) + 1)
    for prec, tol in (("bf16x6", 5e-6), ("f32", 5e-6), ("bf16x3", 1e-4)):
        m.set_precision(prec)
        y = m(x.cuda(), torch.tensor([999, 17]).cuda())
        assert y.shape == g[key + "_int_y"].shape and rel_l2(y, g[key + "_int_y"]) < tol, (key, prec)
        assert rel_l2(m(x.cuda(), torch.tensor([499.5, 20.0]).cuda()), g[key + "_float_y"]) < tol, (key, prec)
    m.set_precision("bf16x6")
    cfg = O.UNetConfig.from_params(params)
    x3 = randn((3, params["in_channels"], 32, 16), 9)                  # other batch / shape, against the oracle run live
    t3 = torch.tensor([3, 500, 998])
    assert rel_l2(m(x3.cuda(), t3.cuda()), O.plain_unet_forward(cfg, sd, x3, t3)) < 5e-6
    from diffusion_models_dsdiff_amd import _lib
    with pytest.raises(_lib.DsdError):
        m(randn((1, params["in_channels"] + 1, 16, 16), 1).cuda(), torch.tensor([1]).cuda())
    with pytest.raises(_lib.DsdError):
        m(randn((1, params["in_channels"], 17, 16), 1).cuda(), torch.tensor([1]).cuda())


def test_latent_pipeline_encode_sample_decode():
    """cond image -> AutoencoderKL.encode -> posterior sample -> 10-step DDIM in latent space with the UNetModel on
    cat([z_t, z_cond]) ('concat' conditioning, ddpm.py:1331-1333) -> decode; every stage against the oracle."""
    from diffusion_models_dsdiff_amd.ldm.models.autoencoder import AutoencoderKL
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    gv = golden("vae")
    dd = json.loads(str(gv["small_cfg"]))
    embed = dd.pop("embed_dim")
    sdv = fixture_params(gv, "small")
    vae = AutoencoderKL(dd, None, embed)
    vae.load_state_dict(sdv, strict=True)
    vc = V.VaeConfig(**dd, embed_dim=embed)
    from diffusion_models_dsdiff_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    up = dict(image_size=8, in_channels=2 * embed, model_channels=32, out_channels=embed, num_res_blocks=1, attention_resolutions=[2],
              channel_mult=[1, 2], num_head_channels=16, legacy=False)
    unet = UNetModel(**up)
    from oracle.synth import synth_params
    sdu = synth_params([(k, tuple(v.shape)) for k, v in unet.state_dict().items()], 55)
    unet.load_state_dict(sdu, strict=True)
    ucfg = O.UNetConfig.from_params(up)

    cond_img = randn((2, 1, 32, 32), 71)
    noise_post = randn((2, embed, 8, 8), 72)
    # encode + posterior sample (scale_factor as in first_stage encoding)
    z_cond = vae.encode(cond_img.cuda()).sample(noise=noise_post.cuda())
    z_cond_o = V.gaussian_sample(V.encode(vc, sdv, cond_img), noise_post)
    assert rel_l2(z_cond, z_cond_o) < 1e-5
    # 10-step DDIM (eta 0) in latent space through the reference call signature; generic callable -> fused update per step
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="10", rescale_timesteps=True, parameterization="v")
    zT = randn((2, embed, 8, 8), 73)
    model = lambda xx, tt, **kw: unet(torch.cat([xx] + kw["c_concat"], 1), tt)
    z0 = d.ddim_sample_loop(model, (2, embed, 8, 8), noise=zT.cuda(), clip_denoised=False, model_kwargs=dict(c_concat=[z_cond]),
                            eta=0.0, device=torch.device("cuda"))
    od = OS.DiffusionA(steps=1000, timestep_respacing="10", rescale_timesteps=True, parameterization="v")
    z0_o = od.ddim_sample_loop(lambda xx, tt: O.plain_unet_forward(ucfg, sdu, xx, tt), zT, torch.zeros((10, 2, embed, 8, 8)),
                               [z_cond_o], clip_denoised=False)
    assert rel_l2(z0, z0_o) < 1e-4
    rec = vae.decode(z0)
    assert rec.shape == (2, 1, 32, 32) and rel_l2(rec, V.decode(vc, sdv, z0_o)) < 1e-4
