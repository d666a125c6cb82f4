#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: DiT-B/8 on 512x512 (4096 tokens, 12 heads of 64, depth 12), one evaluation of a slice batch,
per arithmetic mode, with the per-kernel split from dsd_profile_* (hipEvents on the launch stream).  Synthetic weights.

    python tools/bench_dit.py [--batch 16] [--modes f16,bf16,bf16x6] [--iters 5] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--modes", default="f16,bf16,bf16x6")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--patch", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.DiT_models import DiT
    B = args.batch
    dit = DiT(input_size=args.size, patch_size=args.patch, in_channels=4, hidden_size=args.hidden, depth=args.depth,
              num_heads=args.heads, num_classes=0)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in dit.parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02, generator=g)
            else:
                p.normal_(0.0, 0.02, generator=g)
    x = torch.randn(B, 4, args.size, args.size).cuda()
    t = torch.full((B,), 500.0).cuda()
    res = {"batch": B, "gpu": torch.cuda.get_device_name(0), "tokens": (args.size // args.patch) ** 2, "hidden": args.hidden,
           "depth": args.depth, "heads": args.heads}
    for prec in args.modes.split(","):
        dit.set_precision(prec)
        for _ in range(2):
            out = dit(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(3):
            e0.record()
            for _ in range(args.iters):
                out = dit(x, t)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / args.iters)
        info = dit.plan_info()
        dit.profile(True)
        for _ in range(2):
            dit(x, t)
        rep, runs = dit.profile_report()
        dit.profile(False)
        kinds = {}
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
            per = v["ms"] / runs
            kinds[k] = {"ms": round(per, 3), "calls": v["calls"] // runs,
                        "tflops": round(v["flops"] / runs / per / 1e9, 1) if v["flops"] else None,
                        "gbps": round(v["bytes"] / runs / per / 1e6, 1) if v["bytes"] else None}
        res[prec] = {"ms_per_forward_median": round(sorted(ms)[1], 3), "ms_min": round(min(ms), 3), "ms_max": round(max(ms), 3),
                     "plan_tflop": round(info["flops"] / 1e12, 3), "tflops": round(info["flops"] / sorted(ms)[1] / 1e9, 1),
                     "finite": bool(torch.isfinite(out).all()), "kernels": kinds}
        print(prec, json.dumps(res[prec]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
