#!/usr/bin/env python3
"""bench.py — 256x256 cDDPM slices/sec (1000-step) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 4 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: configs/v2-1-cddpm-ds-disc.yaml U-Net (981.5 M params), 256x256, 1->1
channel (C_in = 2), batch 16 per GPU, 1000-step DDPM (guided-diffusion family, v-parameterisation,
clip_denoised, fixed-large variance), synthetic seeded random-init weights with every zero-initialised
parameter re-randomised (SURVEY.md headline fact 3), Gaussian x_T, clipped-Gaussian condition, on-device
Philox noise.  Inputs are resident in HBM before the timed region.

A "step" is ONE denoising step (network evaluation + fused sampler update) of that chain for the whole
batch: K consecutive steps of the 1000 are timed through dsd_sample(first_step, n_steps) — every step of the
chain runs the same kernels on the same shapes, so  slices/s = B_total / (1000 * t_step).
Slices are independent chains: ranks shard the slice batch, the only collective is the one-off RCCL
broadcast of the packed weights from rank 0 ("scaling": "weak", batch 16 per GPU).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32-in matrix peak
PEAK_HBM_GBS = 8000.0
F_LIVE_PER_SLICE_STEP = 5.5427e12   # algorithmic FLOPs per 256^2 slice per step (dead heads dropped)


def synth_weights_(model, seed):
    """torch.manual_seed(seed) default init (done by the ctor) + N(0, 0.02) for every all-zero parameter, in
    module-tree order."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02, generator=g)


def cpu_baseline(unet_params, sd, H, W, budget_s):
    """The oracle (CPU restatement pinned against the reference) timed on this host's cores: B=1 forwards of the
    same network at the same resolution, as many as fit in ~budget_s (at least 1)."""
    from oracle import unet as O
    cfg = O.UNetConfig.from_params(unet_params)
    # the GPU box reports every host core but a 1-GPU job owns a share of them: use the affinity mask, at most 32 threads
    try:
        n = min(len(os.sched_getaffinity(0)), 32)
    except AttributeError:
        n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    x = torch.randn(1, 2, H, W)
    t = torch.tensor([500])
    times = []
    t_start = time.time()
    while True:
        t0 = time.time()
        O.unet_forward(cfg, sd, x, t)
        times.append(time.time() - t0)
        if time.time() - t_start + times[-1] > budget_s or len(times) >= 5:
            break
    step = sorted(times)[len(times) // 2]
    return {"value": 1.0 / (1000.0 * step), "unit": "slices/s", "cores": n, "kind": "port",
            "sample": f"{len(times)} forward(s) of the same U-Net at {H}x{W}, batch 1, fp32, torch-CPU oracle, "
                      f"median {step:.2f} s/step, extrapolated x1000 steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="slices per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml"))
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel hipEvent pass")
    ap.add_argument("--model-channels", type=int, default=None, help="override (debug only; invalidates the metric)")
    ap.add_argument("--precision", default=os.environ.get("DSD_PRECISION", "bf16x6"), choices=["f32", "bf16x6", "bf16x3", "f16x3"],
                    help="arithmetic of the convolutions (include/dsdiff.h: dsd_set_precision); default = library default")
    ap.add_argument("--no-modes", action="store_true", help="do not also time the other arithmetic modes")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (never set by the driver): run N ranks on ONE device over gloo to exercise the multi-rank code path
    backend = os.environ.get("DSD_BENCH_BACKEND", "nccl")
    if os.environ.get("DSD_BENCH_SINGLE_DEVICE"):
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import yaml
    from diffusion_models_dsdiff_amd import _lib
    from diffusion_models_dsdiff_amd.ldm.util import instantiate_from_config
    from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
    from diffusion_models_dsdiff_amd._sched import run_device_loop

    gpu_name, n_cu, hbm = _lib.require_gpu(local)
    cfg = yaml.safe_load(open(args.config))
    mp = cfg["model"]["params"]
    uc = dict(mp["unet_config"])
    uc["params"] = dict(uc["params"], device_index=local)
    if args.model_channels:
        uc["params"]["model_channels"] = args.model_channels
    torch.manual_seed(2024)
    t0 = time.time()
    model = instantiate_from_config(uc)
    model.set_precision(args.precision)
    synth_weights_(model, 2024)
    n_params = sum(p.numel() for p in model.parameters())
    # one-off weight distribution: rank 0's parameters broadcast as ONE packed blob over RCCL/xGMI
    bcast_ms = None
    if world > 1:
        flat = torch.cat([p.data.reshape(-1) for p in model.parameters()]).to(dev)
        torch.cuda.synchronize()
        tb = time.time()
        dist.broadcast(flat, 0)
        torch.cuda.synchronize()
        bcast_ms = (time.time() - tb) * 1e3
        off = 0
        for p in model.parameters():
            p.data = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
    model.sync_params()
    t_setup = time.time() - t0

    B, H, W = args.batch, args.size, args.size
    diffusion = create_gaussian_diffusion(steps=mp.get("diffusion_steps", 1000), learn_sigma=mp.get("learn_sigma", False),
                                          noise_schedule=mp.get("noise_schedule", "linear"),
                                          predict_xstart=mp.get("predict_xstart", False),
                                          rescale_timesteps=mp.get("rescale_timesteps", False),
                                          timestep_respacing=mp.get("timestep_respacing", ""),
                                          parameterization=mp.get("parameterization", "eps"))
    sched = diffusion._schedule(False, 0.0, bool(mp.get("clip_denoised", True)))
    assert sched.steps == 1000
    g = torch.Generator(device=dev).manual_seed(2025 + rank)
    cond = torch.randn(B, 1, H, W, device=dev, generator=g).clamp_(-1, 1)
    x = torch.randn(B, 1, H, W, device=dev, generator=g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    peak_bf16 = 2500.0

    def measure(precision, steps, warmup, profile_steps):
        """K timed denoising steps in one arithmetic mode (+ optional per-kernel hipEvent pass on rank 0)."""
        nonlocal x
        model.set_precision(precision)
        if warmup > 0:
            x = run_device_loop(model, sched, x, cond, seed=1234, first_step=0, n_steps=warmup)
        else:
            _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))
        barrier()
        t1 = time.time()
        x = run_device_loop(model, sched, x, cond, seed=1234, first_step=warmup, n_steps=steps)
        barrier()
        dt = time.time() - t1
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        ms = dt / steps * 1e3
        info = model.plan_info()
        res = {"precision": precision, "ms_per_step": round(ms, 3), "value": round((B * world) / (1000.0 * ms / 1e3), 6),
               "whole_step_tflops": round(info["flops"] / (ms / 1e3) / 1e12, 2), "finite": bool(torch.isfinite(x).all()),
               "info": info, "roofline": None, "kernels": None}
        if profile_steps > 0 and rank == 0:
            model.profile(True)
            run_device_loop(model, sched, x, cond, seed=1234, first_step=warmup + steps, n_steps=profile_steps)
            rep, runs = model.profile_report()
            model.profile(False)
            tot_ms = sum(v["ms"] for v in rep.values())
            dk, dv = max(rep.items(), key=lambda kv: kv[1]["ms"])
            ach = dv["flops"] / (dv["ms"] / 1e3) / 1e12
            passes = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "f16x3": 3}[precision]
            traffic = None
            # HBM bytes per launch from the committed PMC passes of this same command (tools/pmc_summary.py;
            # separate --pmc runs, FETCH_SIZE x2 on gfx950, KiB -> bytes): counters cannot be read live.
            pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
            if os.path.exists(pmc):
                try:
                    want = {"f32": ("conv_mfma_buf_kernel", "<5>"), "bf16x6": ("conv_split", "<5, 3"),
                            "bf16x3": ("conv_split", "<5, 2, false"), "f16x3": ("conv_split", "<5, 2, true")}[precision]
                    best = 0.0
                    for kname, e in json.load(open(pmc)).items():   # the variant with the most time under PMC
                        if want[0] in kname and want[1] in kname and e.get("total_us_under_pmc", 0) > best:
                            best = e["total_us_under_pmc"]
                            traffic = e.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # achieved = ALGORITHMIC FLOPs (2*MAC of the fp32 convolution) / measured time.  peak = the fp32 matrix peak
            # in every mode ("if bf16x3 is used, still report against fp32 peak and state the bf16 peak"):
            # the split modes issue `passes` bf16 MFMAs per fp32 product, so frac > 1 is possible there.
            res["roofline"] = {"bound": "mfma", "kernel": dk, "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                               "peak_note": "fp32-input MFMA dense peak (157.3); bf16 dense MFMA peak is 2500",
                               "issued_mfma_tflops": round(ach * passes, 1),
                               "issued_frac_of_issued_dtype_peak": round(ach * passes / (PEAK_FP32_MFMA_TFLOPS if passes == 1 else peak_bf16), 4),
                               "launches_per_step": dv["calls"] // max(runs, 1),
                               "avg_launch_ms": round(dv["ms"] / max(dv["calls"], 1), 4),
                               "flops_per_launch": dv["flops"] / max(dv["calls"], 1),
                               "share_of_step_time": round(dv["ms"] / tot_ms, 4),
                               "whole_step_tflops": res["whole_step_tflops"],
                               "whole_step_frac": round(res["whole_step_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)}
            kern = {}
            for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
                e = {"ms_per_step": round(v["ms"] / runs, 3), "calls_per_step": v["calls"] // runs}
                if v["flops"] > 0:
                    e["tflops"] = round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2)
                if v["bytes"] > 0:
                    e["gbs"] = round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1)
                    e["hbm_frac"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
                kern[k] = e
            res["kernels"] = kern
        return res

    main_res = measure(args.precision, args.steps, args.warmup, 0 if args.no_profile else 2)
    ms_per_step, slices_per_s = main_res["ms_per_step"], main_res["value"]
    info, finite, roofline, kernels = main_res["info"], main_res["finite"], main_res["roofline"], main_res["kernels"]
    # the other arithmetic modes, measured in the same process (single-GPU runs only; 2 steps each)
    modes = {args.precision: {k: main_res[k] for k in ("ms_per_step", "value", "whole_step_tflops")}}
    if world == 1 and not args.no_modes:
        for pr in ("f32", "bf16x6", "f16x3", "bf16x3"):
            if pr == args.precision:
                continue
            r = measure(pr, 2, 1, 0 if args.no_profile else 1)
            modes[pr] = {k: r[k] for k in ("ms_per_step", "value", "whole_step_tflops")}
            if r["roofline"]:
                modes[pr]["dominant_kernel"] = {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
        model.set_precision(args.precision)
        # separate line (never part of `value`): the two zero-input streams evaluated once per step instead of per slice
        model.share_zero_streams(True)
        r = measure(args.precision, 2, 1, 0)
        model.share_zero_streams(False)
        modes[args.precision + "+share_zero_streams"] = {
            "ms_per_step": r["ms_per_step"], "value": r["value"], "executed_flops_per_step": r["info"]["flops"],
            "note": "same output up to fp32 rounding; 2 of the 4 encoder streams have all-zero input in the 1->1-channel case and are "
                    "computed at batch 1 (SURVEY.md 7: must be reported separately)"}
        _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))

    # separate line (never part of `value`): the 20-evaluation DPM-Solver++ sampler of the reference
    # (gaussian_diffusion.py:467-522), whole samples end to end through dsd_sample_dpm
    samplers = None
    if world == 1 and not args.no_modes:
        d20 = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True,
                                        parameterization=mp.get("parameterization", "eps"))
        x_T = torch.randn(B, 1, H, W, device=dev, generator=g)
        barrier()
        t1 = time.time()
        y = d20.dpm_solver_sample_loop(model, (B, 1, H, W), model_kwargs=dict(c_concat=[cond]), noise=x_T)
        barrier()
        dt = time.time() - t1
        samplers = {"dpm_solver++_multistep2_20": {
            "network_evaluations": 20, "seconds_per_batch": round(dt, 3), "slices_per_s": round(B / dt, 4),
            "ms_per_evaluation": round(dt / 20 * 1e3, 2), "finite": bool(torch.isfinite(y).all()),
            "note": "logSNR spacing, order 2, dynamic thresholding (radix-select quantile) — a different sampler, not the "
                    "1000-step metric"}}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        cpu = cpu_baseline(uc["params"], sd, H, W, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "256x256 cDDPM slices/sec (1000-step)",
            "value": round(slices_per_s, 6),
            "unit": "slices/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (operands split exactly into 3 bf16 pieces, 6 bf16-MFMA products, f32 accumulate)",
                      "bf16x3": "bf16x3 (2 bf16 pieces per f32 operand, 3 products, f32 accumulate)",
                      "f16x3": "f16x3 (2 fp16 pieces per f32 operand, 3 products, f32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[1]: v2-1-cddpm-ds-disc.yaml U-Net, 256x256 1->1-ch, 1000-step DDPM, "
                                   f"batch {B} per GPU; a step = 1 of the 1000 denoising steps for the whole batch",
                       "slices_per_gpu": B, "image": [H, W], "sampler": "guided-diffusion DDPM, v-param, 1000 steps",
                       "params": n_params, "sharding": f"slices x{world} (no data-path collective)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "modes": modes,
            "samplers": samplers,
            "extra": {"gpu": gpu_name, "compute_units": n_cu, "finite_output": finite,
                      "workspace_GiB": round(info["workspace_bytes"] / 2 ** 30, 2), "launches_per_step": info["launches"],
                      "executed_flops_per_step": info["flops"],
                      "survey_flops_per_step": F_LIVE_PER_SLICE_STEP * B,
                      "seconds_per_1000_step_batch": round(ms_per_step, 3),
                      "setup_s": round(t_setup, 1), "weight_broadcast_ms": bcast_ms},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
