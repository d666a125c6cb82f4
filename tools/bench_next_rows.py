#!/usr/bin/env python3
"""Timings of the rows beyond the headline path (SURVEY f-3 / f-4, BASELINE configs 4 and 5) on one GPU — NOT the bench.py
metric; DESIGN.md §7 quotes them.  Synthetic weights (oracle/synth is not used: torch.randn * 0.02), random inputs.

    python tools/bench_next_rows.py [--batch 16]
"""
import argparse
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fill_(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02, generator=g)
    return mod


def timed(fn, iters, warm=2):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    B = args.batch
    res = {"batch": B, "gpu": torch.cuda.get_device_name(0)}

    # ---- config 4: KL-VAE 256x256 -> 64x64x3 latents (configs/autoencoder_kl_64x64x3.yaml, 1 input channel as the trainer forces)
    from diffusion_models_dsdiff_amd.ldm.models.autoencoder import AutoencoderKL
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "autoencoder_kl_64x64x3.yaml")))["model"]["params"]
    vae = fill_(AutoencoderKL(dict(cfg["ddconfig"]), cfg.get("lossconfig"), cfg["embed_dim"]), 1)
    x = torch.randn(B, 1, 256, 256).cuda()
    ms_e, post = timed(lambda: vae.encode(x), 5)
    z = post.mode()
    ms_d, rec = timed(lambda: vae.decode(z), 5)
    res["vae_256"] = {"encode_ms_per_batch": round(ms_e, 2), "decode_ms_per_batch": round(ms_d, 2),
                      "encode_slices_per_s": round(B / ms_e * 1e3, 1), "decode_slices_per_s": round(B / ms_d * 1e3, 1),
                      "latent": list(z.shape), "finite": bool(torch.isfinite(rec).all())}

    # ---- config 4: the latent-space denoiser (plain UNetModel, 320 channels, on 64x64 latents with concat conditioning)
    from diffusion_models_dsdiff_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    unet = fill_(UNetModel(image_size=64, in_channels=6, out_channels=3, model_channels=320, attention_resolutions=[4, 2, 1],
                           num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_new_attention_order=True), 2)
    zt = torch.randn(B, 6, 64, 64).cuda()
    t = torch.full((B,), 500, dtype=torch.long).cuda()
    ms_u, eps = timed(lambda: unet(zt, t), 5)
    res["latent_unet_64"] = {"ms_per_forward": round(ms_u, 2), "params": sum(p.numel() for p in unet.parameters()),
                             "slices_per_s_1000_steps": round(B / ms_u, 4), "finite": bool(torch.isfinite(eps).all())}

    # ---- config 5: DiT-B/8 on 512x512 (4096 tokens, 12 heads of 64, depth 12)
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.DiT_models import DiT
    dit = fill_(DiT(input_size=512, patch_size=8, in_channels=4, hidden_size=768, depth=12, num_heads=12, num_classes=0), 3)
    xd = torch.randn(B, 4, 512, 512).cuda()
    td = torch.full((B,), 500.0).cuda()
    for prec in ("bf16x6", "f16x3"):
        dit.set_precision(prec)
        ms_t, out = timed(lambda: dit(xd, td), 5)
        res[f"dit_b8_512_{prec}"] = {"ms_per_forward": round(ms_t, 2), "slices_per_s_1000_steps": round(B / ms_t, 4),
                                     "finite": bool(torch.isfinite(out).all())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
