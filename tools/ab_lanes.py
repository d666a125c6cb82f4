#!/usr/bin/env python3
"""Same-process A/B of dsd_set_stream_lanes (the four encoder streams' small levels on four HIP streams) on the headline
network: denoising steps at batch 1, 2, 4, 16, interleaved rounds (sequential / lanes), hipEvents.
    python tools/ab_lanes.py [--batches 1,16] [--rounds 3] [--json out.json]"""
import argparse, json, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_models_dsdiff_amd.ldm.util import instantiate_from_config
from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
from diffusion_models_dsdiff_amd._sched import run_device_loop

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,2,4,16")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--pixels", default="16384")
ap.add_argument("--json", default=None)
args = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml")))
torch.manual_seed(2024)
model = instantiate_from_config(dict(cfg["model"]["params"]["unet_config"]))
g = torch.Generator().manual_seed(2024)
with torch.no_grad():
    for _, p in model.named_parameters():
        if float(p.abs().max()) == 0.0:
            p.normal_(0.0, 0.02, generator=g)
d = create_gaussian_diffusion(steps=1000, parameterization=cfg["model"]["params"].get("parameterization", "eps"))
sched = d._schedule(False, 0.0, True)
out = {"gpu": torch.cuda.get_device_name(0)}
for B in [int(b) for b in args.batches.split(",")]:
    cond = torch.randn(B, 1, 256, 256).cuda().clamp_(-1, 1)
    x = torch.randn(B, 1, 256, 256).cuda()
    steps = 3 if B >= 8 else 10
    variants = [("sequential", False, 0)] + [(f"lanes<= {px} px", True, int(px)) for px in args.pixels.split(",")]
    res = {v[0]: [] for v in variants}
    for r in range(args.rounds):
        for name, on, px in variants:
            model.stream_lanes(on, px)
            run_device_loop(model, sched, x, cond, seed=1, first_step=0, n_steps=1)      # (re)plan + warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            run_device_loop(model, sched, x, cond, seed=1, first_step=1, n_steps=steps)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(round(e0.elapsed_time(e1) / steps, 3))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    out[f"batch_{B}"] = {"ms_per_step": res, "median": med, "ratio_vs_sequential": {k: round(v / med["sequential"], 4) for k, v in med.items()}}
    print(B, json.dumps(out[f"batch_{B}"]), flush=True)
model.stream_lanes(True)
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
