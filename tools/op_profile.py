#!/usr/bin/env python3
"""Per-op hipEvent times of ONE forward of the headline network (sequential launch on one stream), at a given batch:
where a small-batch step spends its time, next to the same op's share of the batch-16 step divided by 16.
    python tools/op_profile.py [--batch 1] [--ref-batch 16] [--top 40] [--json out.json]"""
import argparse, collections, json, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_models_dsdiff_amd.ldm.util import instantiate_from_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--ref-batch", type=int, default=16)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--json", default=None)
ap.add_argument("--dump", default=None, help="write every op of the --batch forward: index, kind, ms, flops, algorithmic bytes")
args = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml")))
torch.manual_seed(2024)
model = instantiate_from_config(dict(cfg["model"]["params"]["unet_config"]))
g = torch.Generator().manual_seed(2024)
with torch.no_grad():
    for _, p in model.named_parameters():
        if float(p.abs().max()) == 0.0:
            p.normal_(0.0, 0.02, generator=g)


def ops_at(B):
    x = torch.randn(B, 2, 256, 256).cuda()
    t = torch.full((B,), 500, device="cuda")
    model(x, t)
    model.profile(True)
    acc = None
    for _ in range(3):
        model(x, t)
        ops = model.profile_ops()
        acc = [list(o) for o in ops] if acc is None else [[a[0], min(a[1], o[1]), a[2], a[3], a[4]] for a, o in zip(acc, ops)]
    model.profile(False)
    return acc


a, r = ops_at(args.batch), ops_at(args.ref_batch)
if args.dump:
    with open(args.dump, "w") as f:
        for i, (k, ms, fl, by, nm) in enumerate(a):
            f.write(f"{i}\t{k}\t{ms * 1e3:.1f}\t{fl:.4g}\t{by:.4g}\t{nm}\n")
scale = args.batch / args.ref_batch
tot_a, tot_r = sum(o[1] for o in a), sum(o[1] for o in r) * scale
print(f"batch {args.batch}: {tot_a:.2f} ms in {len(a)} ops; batch {args.ref_batch} x {scale:g}: {tot_r:.2f} ms in {len(r)} ops")


def grouped(ops, B):
    g_ = collections.OrderedDict()
    for k, ms, fl, by, _nm in ops:
        e = g_.setdefault((k, round(fl / B / 1e6), round(by / B / 1e4)), [0, 0.0])
        e[0] += 1
        e[1] += ms
    return g_


ga, gr = grouped(a, args.batch), grouped(r, args.ref_batch)
rows = []
for key in list(ga) + [k for k in gr if k not in ga]:
    n, ms = ga.get(key, (0, 0.0))
    rn, rms = gr.get(key, (0, 0.0))
    rows.append((key, n, ms, rn, rms * scale))
rows.sort(key=lambda t: -(t[2] - t[4]))
print(f"{'kind':40s} {'MFLOP/smp':>9s} {'n':>4s} {'ms':>8s} {'ref n':>5s} {'ref ms':>8s} {'excess':>8s} {'us/op':>7s}")
for (k, mf, kb), n, ms, rn, rms in rows[:args.top]:
    print(f"{k[:40]:40s} {mf:9d} {n:4d} {ms:8.3f} {rn:5d} {rms:8.3f} {ms - rms:8.3f} {ms / max(n, 1) * 1e3:7.1f}")
kinds = collections.OrderedDict()
for (k, mf, kb), n, ms, rn, rms in rows:
    e = kinds.setdefault(k, [0, 0.0, 0, 0.0])
    e[0] += n; e[1] += ms; e[2] += rn; e[3] += rms
print("per kind:")
for k, (n, ms, rn, rms) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:40]:40s} {'':9s} {n:4d} {ms:8.3f} {rn:5d} {rms:8.3f} {ms - rms:8.3f}")
if args.json:
    json.dump({"batch": args.batch, "ref_batch": args.ref_batch, "total_ms": tot_a, "ref_total_ms_scaled": tot_r,
               "groups": [{"kind": k, "mflop_per_sample": mf, "n": n, "ms": ms, "ref_n": rn, "ref_ms_scaled": rms} for (k, mf, kb), n, ms, rn, rms in rows]},
              open(args.json, "w"), indent=1)
