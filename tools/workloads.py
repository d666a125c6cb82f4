"""BASELINE.json configs[2..4] as measured workloads with their own roofline blocks — imported by bench.py (the `workloads`
key of its JSON line; the headline metric stays configs[1]) and runnable on its own:

    python tools/workloads.py [--batch 16] [--only brats,latent,dit] [--json out.json]

  configs[2]  BraTS: 3 conditioning modalities -> T1ce (x has 4 channels: all four encoder streams live,
              UNet_DS_Diff/model.py:659-663), 50-step DDIM as trainers/trainer_use_gaussian_diff.py:73-76,591-599 configures it
              (timestep_respacing "50", rescale_timesteps, eta 0), 256x256 (the reference pads 240 -> 256), whole samples.
  configs[3]  latent path: KL-VAE of configs/autoencoder_kl_64x64x3.yaml (256x256 -> 64x64x3) encode / decode and one
              evaluation of the latent-space UNetModel (ldm/modules/diffusionmodules/openaimodel.py:571-958).
  configs[4]  DiT-B/8 on 512x512 in fp16 (UNet_DS_Diff/DiT_models.py:101-262 under autocast): one evaluation.

Every block: wall time by hipEvents on the launch stream (median of 3), throughput, and the roofline of the dominant kernel
from a dsd_profile_* pass (hipEvents around every op of the plan): achieved = algorithmic FLOPs x MFMA products per
product / time against the dense peak of the issued dtype, or algorithmic bytes / time against 8 TB/s for a memory-bound
kernel.  `traffic` (PMC HBM bytes per launch) comes from the committed summary profiles/r03_pmc_<workload>.json when it holds
that kernel, else null.  Synthetic weights, random inputs.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_16BIT_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def mfma_products(kind: str) -> float:
    """MFMA products issued per algorithmic product of the op kind (the plan's kind names carry the arithmetic mode)."""
    if kind.startswith(("gemm16", "attention16")):
        return 1.0
    if kind.startswith("conv_wino"):
        return 4.0
    if kind.startswith(("conv_bf16x6", "vae_attention")) or kind == "attention":
        return 6.0
    if kind.startswith(("conv_bf16x3", "conv_f16x3")):
        return 3.0
    return 1.0   # conv_mfma (fp32 MFMA), linears


def issued_peak(kind: str) -> float:
    return PEAK_FP32_MFMA_TFLOPS if kind.startswith(("conv_mfma", "conv_direct", "time_embed", "dit_cond")) else PEAK_16BIT_MFMA_TFLOPS


def kernel_symbol(kind: str):
    """op kind -> the substring its launches carry in a rocprofv3 kernel trace."""
    if kind.startswith("gemm16"):
        return "gemm16_kernel"
    if kind == "attention16":
        return "attention16_"          # attention16_kernel<..> / attention16_dma_kernel<..> (head dim 64)
    if kind == "attention":
        return "attention_split_kernel"
    if kind.startswith("conv_bf16x6"):
        return "conv_split_ad_kernel"
    if kind == "vae_attention":
        return "conv_split"
    if kind.startswith("gn_silu_apply"):
        return "affine_act_kernel"
    return None


def pmc_traffic(workload: str, kind: str):
    """HBM bytes per launch of `kind` from the committed PMC summary of this workload (tools/profile_workloads.sh), or None."""
    path = os.path.join(ROOT, "profiles", f"r03_pmc_{'latent' if workload.startswith('latent') else workload}.json")
    sym = kernel_symbol(kind)
    if not (os.path.exists(path) and sym):
        return None, None
    try:
        d = json.load(open(path))
    except Exception:
        return None, None
    best, out = 0.0, (None, None)
    for name, e in d.items():
        if name == "_meta" or not isinstance(e, dict) or sym not in name.replace(" ", ""):
            continue
        if e.get("total_us_under_pmc", 0) > best and e.get("hbm_bytes_per_launch") is not None:
            best = e["total_us_under_pmc"]
            out = (e["hbm_bytes_per_launch"], {"file": os.path.relpath(path, ROOT), "kernel": name, "launches_under_pmc": e.get("launches"),
                                               "git_head": d.get("_meta", {}).get("git_head")})
    return out


def roofline_block(rep, runs, workload):
    """rep: NativeModule.profile_report() -> the roofline of the kernel that takes the most time + the per-kernel table."""
    tot = sum(v["ms"] for v in rep.values())
    dk, dv = max(rep.items(), key=lambda kv: kv[1]["ms"])
    calls = max(dv["calls"], 1)
    sec = dv["ms"] / 1e3
    traffic, source = pmc_traffic(workload, dk)
    alg_bytes = dv["bytes"] / calls if dv["bytes"] else None
    if dv["flops"] > 0:
        alg = dv["flops"] / sec / 1e12
        prod, peak = mfma_products(dk), issued_peak(dk)
        rl = {"bound": "mfma", "kernel": dk, "achieved": round(alg * prod, 1), "peak": peak, "unit": "TFLOP/s",
              "frac": round(alg * prod / peak, 4), "algorithmic_tflops": round(alg, 1), "mfma_products_per_product": prod,
              "flops_per_launch": dv["flops"] / calls}
    else:
        gbs = dv["bytes"] / sec / 1e9
        rl = {"bound": "hbm", "kernel": dk, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": round(gbs / PEAK_HBM_GBS, 4)}
    rl.update({"traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
               "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic and alg_bytes else None, "traffic_source": source,
               "launches": dv["calls"] // max(runs, 1), "avg_launch_ms": round(dv["ms"] / calls, 4),
               "share_of_time": round(dv["ms"] / tot, 4)})
    kern = {}
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        e = {"ms": round(v["ms"] / runs, 3), "calls": v["calls"] // runs}
        if v["flops"] > 0:
            e["tflops"] = round(v["flops"] / (v["ms"] / 1e3) / 1e12, 1)
            e["frac_issued"] = round(e["tflops"] * mfma_products(k) / issued_peak(k), 4)
        elif v["bytes"] > 0:
            e["gbs"] = round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1)
        kern[k] = e
    return rl, kern


def timed(fn, iters=3, warm=1):
    """median / min / max over `iters` of the hipEvent time of fn() on the current stream."""
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    s = sorted(ms)
    return s[len(s) // 2], s[0], s[-1], out


def profiled(mod, fn, runs=2):
    mod.profile(True)
    for _ in range(runs):
        fn()
    rep, n = mod.profile_report()
    mod.profile(False)
    return rep, n


def fill_(mod, seed, std=0.02):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.normal_(0.0, std, generator=g) if p.dim() > 1 else p.normal_(0.0, 0.02, generator=g)
    return mod


# ------------------------------------------------------------------------------------------------------------ configs[2]
def brats_ddim(model, B, H, W, parameterization="v", seed=2027):
    """model: the DSUnetModel of the headline yaml (already built by bench.py), here fed 4 channels."""
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd._sched import run_device_loop
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(seed)
    cond3 = torch.randn(B, 3, H, W, device=dev, generator=g).clamp_(-1, 1)
    xT = torch.randn(B, 1, H, W, device=dev, generator=g)
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="50", rescale_timesteps=True, parameterization=parameterization)
    run = lambda: d.ddim_sample_loop(model, (B, 1, H, W), noise=xT, clip_denoised=True, model_kwargs=dict(c_concat=[cond3]), eta=0.0)
    med, lo, hi, y = timed(run, iters=1, warm=0)     # one whole 50-step sample of the batch (~20 s): long enough to stand alone
    sched = d._schedule(True, 0.0, True)
    rep, runs = profiled(model, lambda: run_device_loop(model, sched, xT, cond3, seed=1, first_step=10, n_steps=1), runs=2)
    rl, kern = roofline_block(rep, runs, "brats")
    info = model.plan_info()
    return {"workload": f"configs[2]: BraTS 3 modalities -> T1ce (C_in = 4, four live encoder streams), 50-step DDIM (respacing '50', "
                        f"rescaled float timesteps, eta 0), {H}x{W}, batch {B}, whole samples; network = configs/v2-1-cddpm-ds-disc.yaml",
            "metric": "slices/s (50-step DDIM)", "value": round(B / (med / 1e3), 4), "unit": "slices/s",
            "seconds_per_batch": round(med / 1e3, 3), "seconds_min_max": [round(lo / 1e3, 3), round(hi / 1e3, 3)],
            "ms_per_evaluation": round(med / 50, 2), "finite": bool(torch.isfinite(y).all()),
            "executed_flops_per_evaluation": info["flops"], "roofline": rl, "kernels": kern}


# ------------------------------------------------------------------------------------------------------------ configs[3]
def latent_path(B, seed=2028):
    import yaml
    from diffusion_models_dsdiff_amd.ldm.models.autoencoder import AutoencoderKL
    from diffusion_models_dsdiff_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "autoencoder_kl_64x64x3.yaml")))["model"]["params"]
    vae = fill_(AutoencoderKL(dict(cfg["ddconfig"]), cfg.get("lossconfig"), cfg["embed_dim"]), seed)
    x = torch.randn(B, 1, 256, 256).cuda()
    enc_ms, _, _, post = timed(lambda: vae.encode(x))
    z = post.mode()
    dec_ms, _, _, rec = timed(lambda: vae.decode(z))
    out = {"workload": f"configs[3]: latent path — KL-VAE of autoencoder_kl_64x64x3.yaml (256x256 -> {tuple(z.shape[1:])}) + latent-space "
                       f"UNetModel (320 ch, attention at 16/32/64, 6 input channels = latent + concat condition), batch {B}"}
    for name, mod, fn, ms in (("vae_encode", vae._enc, lambda: vae.encode(x), enc_ms),
                              ("vae_decode", vae._dec, lambda: vae.decode(z), dec_ms)):
        rep, runs = profiled(mod, fn)
        rl, kern = roofline_block(rep, runs, "latent_" + name)
        out[name] = {"ms_per_batch": round(ms, 2), "slices_per_s": round(B / ms * 1e3, 1), "roofline": rl, "kernels": kern}
    unet = fill_(UNetModel(image_size=64, in_channels=6, out_channels=3, model_channels=320, attention_resolutions=[4, 2, 1],
                           num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_new_attention_order=True), seed + 1)
    zt = torch.randn(B, 6, 64, 64).cuda()
    t = torch.full((B,), 500, dtype=torch.long).cuda()
    u_ms, _, _, eps = timed(lambda: unet(zt, t))
    rep, runs = profiled(unet, lambda: unet(zt, t))
    rl, kern = roofline_block(rep, runs, "latent_unet")
    out["latent_unet_step"] = {"ms_per_evaluation": round(u_ms, 2), "params": sum(p.numel() for p in unet.parameters()),
                               "executed_flops": unet.plan_info()["flops"], "roofline": rl, "kernels": kern,
                               "finite": bool(torch.isfinite(eps).all())}
    total = enc_ms + 50 * u_ms + dec_ms
    out.update({"metric": "slices/s (encode + 50-step latent DDIM + decode)", "value": round(B / total * 1e3, 3), "unit": "slices/s",
                "finite": bool(torch.isfinite(rec).all())})
    del unet, vae
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------ configs[4]
def dit_eager_proxy(B, dtype=torch.float16):
    """Timing proxy only: the torch ops a DiT-B/8 forward of the reference issues under autocast (DiT_models.py:101-122,224-243 with
    timm's Attention / Mlp: fused-SDPA attention, nn.Linear, LayerNorm without affine, adaLN modulate, tanh-GELU, gated residuals),
    written out with random weights, on this GPU through PyTorch-ROCm eager.  Not the reference (timm is absent here) and not an
    oracle: it exists to put a number on "the same network through the vendor libraries on the same box"."""
    import torch.nn.functional as F
    D, depth, heads, T, p = 768, 12, 12, 4096, 8
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    def w(*s):
        return (torch.randn(*s, device=dev, generator=g) * 0.02)
    blocks = [dict(qkv=(w(3 * D, D), w(3 * D)), proj=(w(D, D), w(D)), fc1=(w(4 * D, D), w(4 * D)), fc2=(w(D, 4 * D), w(D)),
                   ada=(w(6 * D, D), w(6 * D))) for _ in range(depth)]
    pe, fin = (w(D, 4 * p * p), w(D)), (w(p * p * 8, D), w(p * p * 8))
    x = torch.randn(B, 4, 512, 512, device=dev, generator=g)
    c = torch.randn(B, D, device=dev, generator=g)

    def fwd():
        with torch.autocast("cuda", dtype=dtype):
            h = F.linear(x.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B, T, 4 * p * p), *pe)
            for bl in blocks:
                m = F.linear(F.silu(c), *bl["ada"]).chunk(6, dim=1)
                a = F.layer_norm(h, (D,)) * (1 + m[1].unsqueeze(1)) + m[0].unsqueeze(1)
                q, k, v = F.linear(a, *bl["qkv"]).reshape(B, T, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
                a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, D)
                h = h + m[2].unsqueeze(1) * F.linear(a, *bl["proj"])
                a = F.layer_norm(h, (D,)) * (1 + m[4].unsqueeze(1)) + m[3].unsqueeze(1)
                h = h + m[5].unsqueeze(1) * F.linear(F.gelu(F.linear(a, *bl["fc1"]), approximate="tanh"), *bl["fc2"])
            return F.linear(F.layer_norm(h, (D,)), *fin)

    with torch.no_grad():
        ms, lo, hi, y = timed(fwd, iters=3, warm=2)
    out = {"ms_per_evaluation": round(ms, 2), "ms_min_max": [round(lo, 2), round(hi, 2)], "finite": bool(torch.isfinite(y.float()).all()),
           "what": f"the torch ops of a DiT-B/8 forward under torch.autocast({str(dtype)[6:]}) with fused SDPA, random weights, PyTorch-ROCm eager on "
                   f"this GPU (torch {torch.__version__}); a timing proxy, not the reference and not a parity check"}
    del blocks, x
    torch.cuda.empty_cache()
    return out


def dit_step(B, precision="f16", seed=2029):
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.DiT_models import DiT
    dit = fill_(DiT(input_size=512, patch_size=8, in_channels=4, hidden_size=768, depth=12, num_heads=12, num_classes=0), seed)
    dit.set_precision(precision)
    x = torch.randn(B, 4, 512, 512).cuda()
    t = torch.full((B,), 500.0).cuda()
    ms, lo, hi, y = timed(lambda: dit(x, t), iters=5, warm=2)
    rep, runs = profiled(dit, lambda: dit(x, t), runs=3)
    rl, kern = roofline_block(rep, runs, "dit")
    fl = dit.plan_info()["flops"]
    out = {"workload": f"configs[4]: DiT-B/8 on 512x512 (4096 tokens, 12 heads of 64, depth 12), batch {B}, one evaluation, arithmetic "
                       f"{precision} (16-bit operands rounded once, one MFMA per product, fp32 accumulation / softmax / LayerNorm)",
           "metric": "ms per evaluation", "value": round(ms, 3), "unit": "ms", "higher_is_better": False, "dtype": precision,
           "ms_min_max": [round(lo, 3), round(hi, 3)], "slices_per_s_1000_steps": round(B / ms, 3),
           "executed_flops": fl, "whole_step_tflops": round(fl / (ms / 1e3) / 1e12, 1),
           "whole_step_frac_of_16bit_peak": round(fl / (ms / 1e3) / 1e12 / PEAK_16BIT_MFMA_TFLOPS, 4),
           "finite": bool(torch.isfinite(y).all()), "roofline": rl, "kernels": kern,
           "parity": "unpinned by the reference (timm absent); tests/test_half_gpu.py: 3.2e-5 (fp16) / 2.6e-4 (bf16) vs the fp32 oracle"}
    del dit
    torch.cuda.empty_cache()
    try:
        out["torch_eager_same_gpu"] = dit_eager_proxy(B, torch.bfloat16 if precision == "bf16" else torch.float16)
    except Exception as e:          # a side line must never take the workload down
        out["torch_eager_same_gpu"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--only", default="brats,latent,dit")
    ap.add_argument("--dit-precision", default="f16")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    res = {"gpu": torch.cuda.get_device_name(0)}
    only = args.only.split(",")
    if "dit" in only:
        res["dit_b8_512"] = dit_step(args.batch, args.dit_precision)
        print("dit", json.dumps(res["dit_b8_512"]), flush=True)
    if "latent" in only:
        res["latent"] = latent_path(args.batch)
        print("latent", json.dumps(res["latent"]), flush=True)
    if "brats" in only:
        import yaml
        from diffusion_models_dsdiff_amd.ldm.util import instantiate_from_config
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml")))
        torch.manual_seed(2024)
        model = instantiate_from_config(dict(cfg["model"]["params"]["unet_config"]))
        g = torch.Generator().manual_seed(2024)
        with torch.no_grad():
            for _, p in model.named_parameters():
                if float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.02, generator=g)
        res["brats_ddim50"] = brats_ddim(model, args.batch, 256, 256, cfg["model"]["params"].get("parameterization", "eps"))
        print("brats", json.dumps(res["brats_ddim50"]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
