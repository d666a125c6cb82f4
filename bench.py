#!/usr/bin/env python3
"""bench.py — 256x256 cDDPM slices/sec (1000-step) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 4 --warmup 1
    python bench.py --gpus 8 ...            # no WORLD_SIZE in the env: this process launches the 8 ranks itself
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
Slices are independent chains: ranks shard the slice batch, the only collectives are the one-off RCCL
broadcast of the packed weights from rank 0 and the final gather of the samples ("scaling": "weak", batch 16
per GPU); the reference analogue of the broadcast is Disc_diff/guided_diffusion/dist_util.py:54-83.

Rehearsal hooks (never set by the driver): DSD_BENCH_BACKEND=gloo runs the ranks over gloo,
DSD_BENCH_SINGLE_DEVICE=1 puts every rank on device 0, DSD_BENCH_STUB=1 replaces the compute leg by a CPU stub so
that launcher + rendezvous + barrier + max-over-ranks + gather can be tested without a GPU
(tests/test_parallel_cpu.py); a stubbed line says so in `data` and carries no metric.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32-in matrix peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0
F_LIVE_PER_SLICE_STEP = 5.5427e12   # algorithmic FLOPs per 256^2 slice per step (dead heads dropped)
HEADLINE = {"batch": 16, "size": 256, "model_channels": None}
# MFMA products issued per fp32 multiply-add of the convolution, and the matrix peak of the dtype they are issued in
PASSES = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "f16x3": 3}
# The F(2,3)-along-W kernel multiplies 4 transformed values where the direct form multiplies 6: per ALGORITHMIC (direct-form)
# FLOP it issues 2/3 of the MFMA work.  `achieved` must count what the matrix pipes really execute.
WINOGRAD_MFMA_FRACTION = 2.0 / 3.0


def mfma_passes(kernel, precision):
    p = float(PASSES[precision])
    return p * WINOGRAD_MFMA_FRACTION if kernel.startswith("conv_wino") else p
ISSUED_DTYPE = {"f32": "f32", "bf16x6": "bf16", "bf16x3": "bf16", "f16x3": "f16"}


def issued_peak(precision):
    return PEAK_FP32_MFMA_TFLOPS if precision == "f32" else PEAK_16BIT_MFMA_TFLOPS


def headline_args(args):
    return (args.batch == HEADLINE["batch"] and args.size == HEADLINE["size"] and args.model_channels is None
            and os.path.abspath(args.config) == os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=3, help="the K timed steps are run this many times; value = median repeat")
    ap.add_argument("--batch", type=int, default=16, help="slices per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml"))
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-eager-ref", action="store_true", help="skip the oracle-on-this-GPU (PyTorch-ROCm eager) line of the baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel hipEvent pass")
    ap.add_argument("--model-channels", type=int, default=None, help="override (debug only; invalidates the metric)")
    ap.add_argument("--precision", default=os.environ.get("DSD_PRECISION", "bf16x6"), choices=list(PASSES),
                    help="arithmetic of the convolutions (include/dsdiff.h: dsd_set_precision); default = library default")
    ap.add_argument("--no-modes", action="store_true", help="do not also time the other arithmetic modes")
    ap.add_argument("--no-workloads", action="store_true", help="skip the BASELINE configs[2..4] lines (BraTS DDIM, latent path, DiT)")
    ap.add_argument("--graph", action="store_true", help="replay the captured hipGraph of a step instead of launching every "
                                                         "kernel from the host (measured equal at batch 16, no gain at batch 1)")
    ap.add_argument("--winograd", action="store_true", help="bf16x6 only: the F(2,3)-along-W kernel for the large 3x3 layers "
                                                            "(include/dsdiff.h: dsd_set_winograd); otherwise a separate line in `modes`")
    return ap.parse_args(argv)


# ======================================================================================== launcher (parent process)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks (one per GPU) through
    torch.distributed.run, relay rank 0's JSON line, exit with the worst child code.  This process never touches HIP
    (no torch.cuda call before or after), so the children own the devices."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DSD_BENCH_LAUNCHED_BY_PARENT"] = "1"
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:                      # stderr passes straight through
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
    return rc


# ======================================================================================== compute legs
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
    same network at the same resolution on a FIXED seeded input, as many as fit in ~budget_s (at least 1).  The
    output of that forward is kept: the caller runs the same input through the GPU path (rel_l2_vs_cpu)."""
    from oracle import unet as O
    from oracle.synth import randn
    cfg = O.UNetConfig.from_params(unet_params)
    # Cores: everything this job may run on — the affinity mask, cut to the cgroup CPU quota when the container has one (a
    # 1-GPU job on the GPU box sees all 256 host threads in its mask but owns a share of them: 256 oversubscribed threads
    # measured 258 s per forward where 32 take 5.5 s).  Where neither bound is informative the thread count is PROBED: one
    # 64x64 forward per candidate, the fastest wins.  `cores` in the report = the threads actually used.
    try:
        n_aff = len(os.sched_getaffinity(0))
    except AttributeError:
        n_aff = os.cpu_count() or 1
    quota = cgroup_cpu_quota()
    probe = None
    if quota is not None:
        n = max(1, min(n_aff, int(quota + 0.5)))
    elif n_aff <= 32:
        n = n_aff
    else:
        probe = {}
        xp, tp = randn((1, 2, 64, 64), 8), torch.tensor([500])
        for cand in sorted({c for c in (16, 32, 64, 128, n_aff) if c <= n_aff}):
            torch.set_num_threads(cand)
            O.unet_forward(cfg, sd, xp, tp)                      # (first call at this thread count: pool start-up)
            t0 = time.time()
            O.unet_forward(cfg, sd, xp, tp)
            probe[cand] = round(time.time() - t0, 3)
            if probe[cand] > 3.0 * min(probe.values()):
                break                                            # clearly past the knee: do not try more threads
        n = min(probe, key=probe.get)
    torch.set_num_threads(n)
    x = randn((1, 2, H, W), 7)
    t = torch.tensor([500])
    times, y = [], None
    t_start = time.time()
    while True:
        t0 = time.time()
        y = O.unet_forward(cfg, sd, x, t)[0]
        times.append(time.time() - t0)
        if time.time() - t_start + times[-1] > budget_s or len(times) >= 5:
            break
    step = sorted(times)[len(times) // 2]
    rep = {"value": 1.0 / (1000.0 * step), "unit": "slices/s", "cores": n, "cpu_model": cpu_model(), "host_cores_total": os.cpu_count(),
           "affinity_cores": n_aff, "cgroup_cpu_quota": quota, "thread_probe_s_per_64x64_forward": probe, "kind": "port",
           "sample": f"{len(times)} forward(s) of the same U-Net at {H}x{W}, batch 1, fp32, torch-CPU oracle, "
                     f"median {step:.2f} s/step, extrapolated x1000 steps"}
    return rep, x, t, y


def eager_gpu_reference(unet_params, sd, B, H, W, dev):
    """Part of the baseline leg (rank 0, N = 1): the SAME oracle — a restatement of the reference's forward in plain torch ops —
    run on this box's GPU through PyTorch-ROCm eager (MIOpen / rocBLAS, fp32), i.e. what the reference's own code path costs on
    an MI355X.  One warm-up and two timed forwards at the headline batch; reported next to the CPU figure, never part of `value`,
    never on a product path."""
    from oracle import unet as O
    cfg = O.UNetConfig.from_params(unet_params)
    sdg = {k: v.to(dev) for k, v in sd.items()}
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.randn(B, 2, H, W, device=dev, generator=g)
    t = torch.full((B,), 500, device=dev)
    with torch.no_grad():
        O.unet_forward(cfg, sdg, x, t)                # MIOpen picks its kernels on the first call per shape
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            O.unet_forward(cfg, sdg, x, t)
            e1.record()
            torch.cuda.synchronize(dev)
            ts.append(e0.elapsed_time(e1))
    del sdg
    torch.cuda.empty_cache()
    ms = min(ts)
    return {"ms_per_step": round(ms, 2), "value": round(B / (1000.0 * ms / 1e3), 6), "unit": "slices/s",
            "what": f"the fp32 oracle (plain torch ops, the reference's own code path) on this GPU through PyTorch-ROCm eager, batch {B}, "
                    f"{H}x{W}, best of 2 forwards after a warm-up; torch {torch.__version__}"}


def cgroup_cpu_quota():
    """CPUs this container may use per the cgroup controller (v2 cpu.max, v1 cfs quota / period), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def kernel_symbol(kind):
    """Plan kind (dsd_profile_get, e.g. "conv_bf16x6<5>", "conv_bf16x3<4>/r128+splitk", "conv_mfma<5>") -> the kernel symbol
    the rocprofv3 summaries list it under (spaces removed), or None."""
    m = re.match(r"conv_(bf16x6|bf16x3|f16x3|mfma)<(\d)>(/r128|/staged)?", kind)
    if kind.startswith("conv_wino"):
        return "conv_wino_kernel"
    if not m:
        return {"gn_silu_apply": "affine_act_kernel<1>", "gn_apply": "affine_act_kernel<0>", "gn_stats": "gn_stats_kernel",
                "attention": "attention"}.get(kind)
    mode, nt, st = m.group(1), m.group(2), m.group(3)
    if mode == "mfma":
        return f"conv_mfma_buf_kernel<{nt}>"
    np_, f16 = {"bf16x6": ("3", "false"), "bf16x3": ("2", "false"), "f16x3": ("2", "true")}[mode]
    if st == "/staged":
        return f"conv_split_kernel<{nt},{np_},{f16}>"
    # (the product instantiations carry four more template arguments: LDS-DMA weights off, diagnostics off, tap reuse and
    # fused GroupNorm apply on / off)
    tr, gn = ("/tr" in kind), kind.endswith("+gn")
    return f"conv_split_ad_kernel<{nt},{np_},{f16},{'1' if st == '/r128' else '2'},false,0,{'true' if tr else 'false'},{'true' if gn else 'false'}>"


def latest_pmc(precision="bf16x6"):
    """Newest committed PMC summary of this arithmetic mode (profiles/rNN_pmc[_<mode>].json, tools/profile_round.sh)."""
    best, bn = None, -1
    suffix = "" if precision == "bf16x6" else "_" + precision
    for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")):
        m = re.match(r"r(\d+)_pmc" + re.escape(suffix) + r"\.json$", os.path.basename(p))
        if m and int(m.group(1)) > bn:
            best, bn = p, int(m.group(1))
    return best


class StubLeg:
    """CPU stand-in for the compute leg (DSD_BENCH_STUB=1): exercises everything in this file that is not the GPU."""
    stub = True

    def __init__(self, args, rank, local, world):
        self.args, self.rank, self.world = args, rank, world
        self.dev = torch.device("cpu")
        self.x = torch.full((args.batch, 1, 8, 8), float(rank))
        self.gpu_name, self.n_cu, self.n_params, self.t_setup, self.bcast_ms = "stub", 0, 0, 0.0, None
        self.bcast_buckets = None
        if world > 1:
            # the same bucketed weight broadcast the GPU leg runs, on a small fake parameter table: only rank 0 holds values
            from diffusion_models_dsdiff_amd import parallel
            named = [("a.weight", (1000,)), ("b.weight", (3, 1000)), ("c.bias", (10,)), ("d.weight", (7, 11))]
            src = {nm: torch.arange(parallel._numel(sh), dtype=torch.float32).reshape(sh) + i for i, (nm, sh) in enumerate(named)}
            got = {}
            t0 = time.time()
            self.bcast_buckets = parallel.broadcast_params_bucketed(named, (lambda nm: src[nm]) if rank == 0 else None,
                                                                    lambda nm, t: got.__setitem__(nm, t.clone()), 0, None, 2048)
            self.bcast_ms = (time.time() - t0) * 1e3
            assert self.bcast_buckets == 3 and all(torch.equal(got[nm], src[nm]) for nm, _ in named)

    def sync(self):
        pass

    def measure(self, precision, steps, warmup, profile_steps, barrier, reduce_max, repeats=1):
        reps = []
        for _ in range(max(1, repeats)):
            barrier()
            t1 = time.time()
            for _ in range(steps):
                time.sleep(0.002 * (1 + self.rank))
            barrier()
            reps.append(reduce_max((time.time() - t1) / steps * 1e3))
        ms, per_rank = sorted(reps, key=lambda r: r[0])[len(reps) // 2]
        return {"precision": precision, "ms_per_step": round(ms, 3), "per_rank_ms_per_step": per_rank,
                "timing": {"method": "time.time() around K sleeps (stub)", "repeats": len(reps),
                           "ms_per_step_repeats": [round(r[0], 3) for r in reps]},
                "value": round((self.args.batch * self.world) / (1000.0 * ms / 1e3), 6), "whole_step_tflops": 0.0,
                "finite": True, "info": {"workspace_bytes": 0, "launches": 0, "flops": 0.0}, "roofline": None,
                "kernels": None}


class GpuLeg:
    stub = False

    def __init__(self, args, rank, local, world):
        import yaml
        from diffusion_models_dsdiff_amd import _lib
        from diffusion_models_dsdiff_amd.ldm.util import instantiate_from_config
        from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
        self.args, self.rank, self.world = args, rank, world
        torch.cuda.set_device(local)
        self.dev = dev = torch.device("cuda", local)
        self.gpu_name, self.n_cu, _ = _lib.require_gpu(local)
        cfg = yaml.safe_load(open(args.config))
        self.mp = mp = cfg["model"]["params"]
        uc = dict(mp["unet_config"])
        uc["params"] = dict(uc["params"], device_index=local)
        if args.model_channels:
            uc["params"]["model_channels"] = args.model_channels
        self.unet_params = uc["params"]
        from diffusion_models_dsdiff_amd import parallel
        torch.manual_seed(2024)
        t0 = time.time()
        # only rank 0 initialises the 981.5 M parameters on its host; the others allocate storage and receive the values
        if rank == 0:
            self.model = model = instantiate_from_config(uc)
            synth_weights_(model, 2024)
        else:
            with parallel.empty_init():
                self.model = model = instantiate_from_config(uc)
        model.set_precision(args.precision)
        model.use_graph(args.graph)
        model.winograd(args.winograd)
        self.n_params = sum(p.numel() for p in model.parameters())
        # one-off weight distribution (the analogue of Disc_diff/guided_diffusion/dist_util.py:54-83): rank 0's parameters in
        # 256 MB buckets over RCCL/xGMI, every bucket uploaded into the library's slab straight from the receive buffer
        self.bcast_ms, self.bcast_buckets = None, None
        if world > 1:
            named = [(nm, tuple(p.shape)) for nm, p in model.named_parameters()]
            src = dict(model.named_parameters()) if rank == 0 else None
            torch.cuda.synchronize()
            tb = time.time()
            self.bcast_buckets = parallel.broadcast_params_bucketed(named, (lambda nm: src[nm]) if rank == 0 else None,
                                                                    model.upload_param, 0, dev)
            torch.cuda.synchronize()
            self.bcast_ms = (time.time() - tb) * 1e3
        else:
            model.sync_params(force=True)
        self.t_setup = time.time() - t0
        B, H, W = args.batch, args.size, args.size
        diffusion = create_gaussian_diffusion(steps=mp.get("diffusion_steps", 1000), learn_sigma=mp.get("learn_sigma", False),
                                              noise_schedule=mp.get("noise_schedule", "linear"),
                                              predict_xstart=mp.get("predict_xstart", False),
                                              rescale_timesteps=mp.get("rescale_timesteps", False),
                                              timestep_respacing=mp.get("timestep_respacing", ""),
                                              parameterization=mp.get("parameterization", "eps"))
        self.sched = diffusion._schedule(False, 0.0, bool(mp.get("clip_denoised", True)))
        assert self.sched.steps == 1000
        self.g = g = torch.Generator(device=dev).manual_seed(2025 + rank)
        self.cond = torch.randn(B, 1, H, W, device=dev, generator=g).clamp_(-1, 1)
        self.x = torch.randn(B, 1, H, W, device=dev, generator=g)

    def sync(self):
        torch.cuda.synchronize()

    def measure(self, precision, steps, warmup, profile_steps, barrier, reduce_max, repeats=1):
        """K timed denoising steps in one arithmetic mode, `repeats` times (+ optional per-kernel hipEvent pass on rank 0).
        Every repeat is bracketed by barrier + synchronize on both sides and timed twice: by hipEvents recorded on the
        stream the steps are launched on (torch's current stream = the one handed to dsd_sample) and by the host clock;
        per repeat the MAX over ranks counts, the reported step time is the MEDIAN repeat of the event timing
        (boxes differ by +-4 %, launches are asynchronous)."""
        from diffusion_models_dsdiff_amd import _lib
        from diffusion_models_dsdiff_amd._sched import run_device_loop
        model, sched, cond = self.model, self.sched, self.cond
        B, H, W = self.args.batch, self.args.size, self.args.size
        model.set_precision(precision)
        if warmup > 0:
            self.x = run_device_loop(model, sched, self.x, cond, seed=1234, first_step=0, n_steps=warmup)
        else:
            _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))
        ev_reps, wall_reps, first = [], [], warmup
        for _ in range(max(1, repeats)):
            if first + steps > sched.steps:
                first = warmup                     # every step of the chain runs the same kernels on the same shapes
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t1 = time.time()
            e0.record()
            self.x = run_device_loop(model, sched, self.x, cond, seed=1234, first_step=first, n_steps=steps)
            e1.record()
            barrier()
            dt = time.time() - t1
            ev_reps.append(reduce_max(e0.elapsed_time(e1) / steps))
            wall_reps.append(reduce_max(dt / steps * 1e3))
            first += steps
        ms, per_rank = sorted(ev_reps, key=lambda r: r[0])[len(ev_reps) // 2]
        info = model.plan_info()
        res = {"precision": precision, "ms_per_step": round(ms, 3), "per_rank_ms_per_step": per_rank,
               "timing": {"method": "hipEvents on the launch stream around K steps, between barrier + synchronize; max over ranks "
                                    "per repeat, median over repeats", "repeats": len(ev_reps),
                          "ms_per_step_repeats": [round(r[0], 3) for r in ev_reps],
                          "ms_per_step_min": round(min(r[0] for r in ev_reps), 3), "ms_per_step_max": round(max(r[0] for r in ev_reps), 3),
                          "host_clock_ms_per_step_repeats": [round(r[0], 3) for r in wall_reps]},
               "value": round((B * self.world) / (1000.0 * ms / 1e3), 6),
               "whole_step_tflops": round(info["flops"] / (ms / 1e3) / 1e12, 2), "finite": bool(torch.isfinite(self.x).all()),
               "info": info, "roofline": None, "kernels": None}
        if profile_steps > 0 and self.rank == 0:
            res["roofline"], res["kernels"] = self.profile(precision, warmup + steps, profile_steps, res)
        return res

    def sustained_mfma(self, achieved):
        """The issued-bf16 rate THIS device holds on bare v_mfma_f32_32x32x16_bf16 loops with the dominant kernel's accumulator
        tile on random operands (csrc/peak.hip), measured in this run right after the network: `peak` above is the nominal
        2.4 GHz figure, which no MFMA-dense loop reaches because the chip lowers its clock under that load."""
        import ctypes as C
        from diffusion_models_dsdiff_amd import _lib
        out = {"unit": "TFLOP/s issued bf16", "how": "dsd_bench_mfma_peak: 2048 workgroups, ~20 ms launches, mean of 5"}
        for name, variant in (("mfma_only_registers", 0), ("mfma_plus_lds_weight_fragments", 1), ("mfma_only_zero_operands", 2)):
            ms, tf = C.c_float(), C.c_double()
            _lib.check(_lib.lib().dsd_bench_mfma_peak(variant, 8, 20.0, 5, C.byref(ms), C.byref(tf)))
            out[name] = round(tf.value, 1)
        out["frac_of_mfma_only"] = round(achieved / out["mfma_only_registers"], 4)
        out["frac_of_mfma_plus_lds"] = round(achieved / out["mfma_plus_lds_weight_fragments"], 4)
        return out

    def profile(self, precision, first_step, profile_steps, res):
        """Per-kernel durations measured live with hipEvents on the launch stream (dsd_profile_*; the captured graph is
        bypassed while profiling), and the roofline block of the dominant kernel."""
        from diffusion_models_dsdiff_amd._sched import run_device_loop
        model = self.model
        model.profile(True)
        run_device_loop(model, self.sched, self.x, self.cond, seed=1234, first_step=first_step, n_steps=profile_steps)
        rep, runs = model.profile_report()
        model.profile(False)
        tot_ms = sum(v["ms"] for v in rep.values())
        dk, dv = max(rep.items(), key=lambda kv: kv[1]["ms"])
        calls = max(dv["calls"], 1)
        alg = dv["flops"] / (dv["ms"] / 1e3) / 1e12          # algorithmic (fp32-equivalent 2*MAC) TFLOP/s of its launches
        passes = mfma_passes(dk, precision)
        peak = issued_peak(precision)
        # HBM bytes per launch from the newest committed PMC passes of this same command (tools/profile_round.sh; separate
        # --pmc runs, FETCH_SIZE x2 on gfx950, KiB -> bytes): counters cannot be read live, so the figure carries its
        # provenance and is dropped when the summary does not contain the kernel this run found dominant.
        traffic, source = None, None
        pmc = latest_pmc(precision)
        if pmc:
            try:
                d = json.load(open(pmc))
                meta = d.get("_meta", {})
                want = kernel_symbol(dk)
                best = 0.0
                for kname, e in d.items():
                    if kname == "_meta" or not isinstance(e, dict) or want is None:
                        continue
                    if want in kname.replace(" ", "") and e.get("total_us_under_pmc", 0) > best \
                            and meta.get("precision", "bf16x6") == precision:
                        best = e["total_us_under_pmc"]
                        traffic = e.get("hbm_bytes_per_launch")
                        source = {"file": os.path.relpath(pmc, ROOT), "kernel": kname, "git_head": meta.get("git_head"),
                                  "command": meta.get("command"), "launches_under_pmc": e.get("launches"),
                                  "mfma_pipe_util": e.get("mfma_pipe_util"), "effective_clock_GHz": e.get("effective_clock_GHz")}
            except Exception:
                traffic, source = None, None
        alg_bytes = dv["bytes"] / calls
        roofline = {
            "bound": "mfma", "kernel": dk,
            # achieved / peak / frac are PHYSICAL: MFMA FLOPs issued in the dtype the matrix cores run (every fp32
            # product = `passes` bf16/f16 MFMA products, each counted as 2*MAC) against that dtype's dense peak
            "achieved": round(alg * passes, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(alg * passes / peak, 4),
            "issued_dtype": ISSUED_DTYPE[precision], "mfma_products_per_fp32_product": passes,
            "algorithmic_tflops": round(alg, 2),                        # 2*MAC of the fp32 convolution / time
            "frac_vs_fp32_mfma_peak": round(alg / PEAK_FP32_MFMA_TFLOPS, 4),   # SURVEY 8d yard-stick (can exceed 1)
            "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
            "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic and alg_bytes else None,
            "traffic_source": source,
            "launches_per_step": dv["calls"] // max(runs, 1), "avg_launch_ms": round(dv["ms"] / calls, 4),
            "flops_per_launch": dv["flops"] / calls, "share_of_step_time": round(dv["ms"] / tot_ms, 4),
            "whole_step_algorithmic_tflops": res["whole_step_tflops"],
            "whole_step_frac_issued": round(sum(v["flops"] * mfma_passes(k, precision) for k, v in rep.items())
                                            / (tot_ms / 1e3) / 1e12 / peak, 4),
            "whole_step_frac_vs_fp32_mfma_peak": round(res["whole_step_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)}
        if ISSUED_DTYPE[precision] == "bf16":
            roofline["sustained"] = self.sustained_mfma(alg * passes)
        kern = {}
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
            e = {"ms_per_step": round(v["ms"] / runs, 3), "calls_per_step": v["calls"] // runs}
            if v["flops"] > 0:
                e["tflops"] = round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2)
            if v["bytes"] > 0:
                e["gbs"] = round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1)
                e["hbm_frac"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
            kern[k] = e
        return roofline, kern


# ======================================================================================== one rank
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"
    backend = os.environ.get("DSD_BENCH_BACKEND", "nccl")
    stub = bool(os.environ.get("DSD_BENCH_STUB"))
    if os.environ.get("DSD_BENCH_SINGLE_DEVICE"):
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    leg = (StubLeg if stub else GpuLeg)(args, rank, local, world)
    dev = leg.dev

    def barrier():
        if world > 1:
            dist.barrier()
        leg.sync()

    def reduce_max(ms):
        """MAX over ranks of the per-step time (the job is as slow as its slowest rank) + every rank's own figure."""
        if world == 1:
            return ms, [round(ms, 3)]
        mine = torch.tensor([ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        vals = [float(v.item()) for v in allr]
        return max(vals), [round(v, 3) for v in vals]

    main_res = leg.measure(args.precision, args.steps, args.warmup, 0 if args.no_profile else 2, barrier, reduce_max,
                           repeats=args.repeats)
    # per-rank device footprint (slab + arena + weight pieces + scratch, dsd_device_bytes): no rank may hold a second copy
    dev_bytes = [main_res["info"].get("device_bytes", 0)]
    if world > 1:
        mine = torch.tensor([float(dev_bytes[0])], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dev_bytes = [int(v.item()) for v in allr]
        if not stub:
            assert max(dev_bytes) - min(dev_bytes) <= 0.01 * max(dev_bytes), f"ranks differ in device footprint: {dev_bytes}"
    ms_per_step, slices_per_s = main_res["ms_per_step"], main_res["value"]
    info, finite, roofline, kernels = main_res["info"], main_res["finite"], main_res["roofline"], main_res["kernels"]

    # end of the job: the samples go to rank 0 (256 KB per slice) — the only other collective on the path
    gather_ms = None
    if world > 1:
        barrier()
        tg = time.time()
        parts = [torch.empty_like(leg.x) for _ in range(world)] if rank == 0 else None
        dist.gather(leg.x, parts, 0)
        leg.sync()
        gather_ms = (time.time() - tg) * 1e3
        if rank == 0 and stub:
            assert [float(p.flatten()[0]) for p in parts] == [float(r) for r in range(world)]
    rccl_ranks = dist.get_world_size() if world > 1 else 1

    workloads = None
    modes, samplers, cpu, rel = {args.precision: {k: main_res[k] for k in ("ms_per_step", "value", "whole_step_tflops")}}, None, None, None
    if not stub and world == 1:
        from diffusion_models_dsdiff_amd import _lib
        from diffusion_models_dsdiff_amd.Disc_diff.guided_diffusion.script_util import create_gaussian_diffusion
        model, B, H, W = leg.model, args.batch, args.size, args.size
        if not args.no_modes:
            # the other arithmetic modes, measured in the same process (2 steps each)
            for pr in ("f32", "bf16x6", "f16x3", "bf16x3"):
                if pr == args.precision:
                    continue
                r = leg.measure(pr, 2, 1, 0 if args.no_profile else 1, barrier, reduce_max)
                modes[pr] = {k: r[k] for k in ("ms_per_step", "value", "whole_step_tflops")}
                if r["roofline"]:
                    modes[pr]["dominant_kernel"] = {k: r["roofline"][k] for k in ("kernel", "achieved", "peak", "frac",
                                                                                  "algorithmic_tflops", "avg_launch_ms")}
            model.set_precision(args.precision)
            # separate line (never part of `value`): the two zero-input streams evaluated once per step instead of per slice
            model.share_zero_streams(True)
            r = leg.measure(args.precision, 2, 1, 0, barrier, reduce_max)
            model.share_zero_streams(False)
            modes[args.precision + "+share_zero_streams"] = {
                "ms_per_step": r["ms_per_step"], "value": r["value"], "executed_flops_per_step": r["info"]["flops"],
                "note": "same output up to fp32 rounding; 2 of the 4 encoder streams have all-zero input in the 1->1-channel case and are "
                        "computed at batch 1 (SURVEY.md 7: must be reported separately)"}
            # separate line: the large 3x3 layers on the F(2,3)-along-W kernel (1.5x fewer MFMAs; conv_wino.hip)
            if not args.winograd and args.precision == "bf16x6":
                model.winograd(True)
                r = leg.measure(args.precision, 3, 1, 0 if args.no_profile else 1, barrier, reduce_max)
                model.winograd(False)
                modes["bf16x6+winograd"] = {k: r[k] for k in ("ms_per_step", "value", "whole_step_tflops")}
                if r["roofline"]:
                    modes["bf16x6+winograd"]["dominant_kernel"] = {k: r["roofline"][k] for k in (
                        "kernel", "achieved", "peak", "frac", "algorithmic_tflops", "avg_launch_ms", "mfma_products_per_fp32_product")}
            # separate line: the tap-reuse kernel's matrix work issued as v_mfma_f32_16x16x32_bf16 (conv_tr16.hip),
            # interleaved with the default shape in this process: A B A B, 3 steps each
            if args.precision == "bf16x6":
                ab = {0: [], 1: []}
                for rnd in range(2):
                    for on in (0, 1):
                        _lib.lib().dsd_set_conv_mfma16(on)
                        r = leg.measure(args.precision, 3, 1, 0, barrier, reduce_max)
                        ab[on].append(r["ms_per_step"])
                _lib.lib().dsd_set_conv_mfma16(0)
                modes["bf16x6+mfma16x16x32"] = {"ms_per_step": round(min(ab[1]), 3), "value": round(B / (1000.0 * min(ab[1]) / 1e3), 6),
                                               "ab_ms_per_step": {"32x32x16": ab[0], "16x16x32": ab[1]},
                                               "note": "same-process interleaved A/B of the dominant kernel's MFMA shape; csrc/peak.hip bare loops: "
                                                       "16x16x32 holds 1.13x (registers) / 1.06x (LDS-fed) the 32x32x16 rate on random data"}
            # separate line: the network evaluation of a step replayed as ONE captured hipGraph instead of ~800 host launches
            if not args.graph:
                model.use_graph(True)
                r = leg.measure(args.precision, 3, 2, 0, barrier, reduce_max)
                model.use_graph(False)
                modes[args.precision + "+hip_graph"] = {"ms_per_step": r["ms_per_step"], "value": r["value"],
                                                         "graph": model.graph_stats()}
            _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))
            # separate line (never part of `value`): the 20-evaluation DPM-Solver++ sampler of the reference
            # (gaussian_diffusion.py:467-522), whole samples end to end through dsd_sample_dpm
            d20 = create_gaussian_diffusion(steps=1000, timestep_respacing="20", rescale_timesteps=True,
                                            parameterization=leg.mp.get("parameterization", "eps"))
            x_T = torch.randn(B, 1, H, W, device=dev, generator=leg.g)
            barrier()
            t1 = time.time()
            y = d20.dpm_solver_sample_loop(model, (B, 1, H, W), model_kwargs=dict(c_concat=[leg.cond]), noise=x_T)
            barrier()
            dt = time.time() - t1
            samplers = {"dpm_solver++_multistep2_20": {
                "network_evaluations": 20, "seconds_per_batch": round(dt, 3), "slices_per_s": round(B / dt, 4),
                "ms_per_evaluation": round(dt / 20 * 1e3, 2), "finite": bool(torch.isfinite(y).all()),
                "note": "logSNR spacing, order 2, dynamic thresholding (radix-select quantile) — a different sampler, not the "
                        "1000-step metric"}}
        if not args.no_workloads and headline_args(args):
            # BASELINE configs[2..4], each with its own roofline block (tools/workloads.py); never part of `value`
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import workloads as WL
            workloads = {}
            try:
                model.set_precision(args.precision)
                workloads["brats_ddim50"] = WL.brats_ddim(model, B, H, W, leg.mp.get("parameterization", "eps"))
                _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))
            except Exception as e:          # a side line must never take the headline down
                workloads["brats_ddim50"] = {"error": repr(e)}
        if args.cpu_seconds > 0:
            # CPU leg: the oracle timed on this host AND used as the checker of the GPU path on the same input, same run
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            cpu, x1, t1_, y_cpu = cpu_baseline(leg.unet_params, sd, H, W, args.cpu_seconds)
            if not args.no_eager_ref and headline_args(args):
                try:
                    cpu["torch_eager_same_gpu"] = eager_gpu_reference(leg.unet_params, sd, B, H, W, dev)
                except Exception as e:          # a side line must never take the headline down
                    cpu["torch_eager_same_gpu"] = {"error": repr(e)}
            model.set_precision(args.precision)
            y_gpu = model._run(x1.to(dev), t1_.to(dev), want_feats=False)[0].cpu()
            err = float((y_gpu.double() - y_cpu.double()).norm() / y_cpu.double().norm())
            rel = {f"forward_{H}": err, "mode": args.precision, "tolerance": 1e-4,
                   "what": f"one forward of the same network at {H}x{W}, batch 1, seeded input, t = 500: GPU path ({args.precision}) "
                           "vs the fp32 CPU oracle whose timing is cpu_baseline; tests/test_model_gpu.py::test_full_size_vs_oracle "
                           "adds the last steps of the 1000-step chain and a row of a batch-16 forward"}
            _lib.check(_lib.lib().dsd_plan(model._h, B, 2, H, W))

    if workloads is not None:
        # the other networks need the device memory the 981.5 M model holds (slab + pieces + 26 GiB arena): run them last
        del model
        leg.model = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        for key, fn in (("latent", lambda: WL.latent_path(args.batch)), ("dit_b8_512", lambda: WL.dit_step(args.batch, "f16"))):
            try:
                workloads[key] = fn()
            except Exception as e:
                workloads[key] = {"error": repr(e)}
    if rank == 0:
        B, H, W = args.batch, args.size, args.size
        headline = (B == HEADLINE["batch"] and H == HEADLINE["size"] and args.model_channels is None and not stub
                    and os.path.abspath(args.config) == os.path.join(ROOT, "configs", "v2-1-cddpm-ds-disc.yaml"))
        net = "v2-1-cddpm-ds-disc.yaml U-Net" + (f" with model_channels={args.model_channels}" if args.model_channels else "")
        workload = (f"{'configs[1]: ' if headline else 'DEBUG OVERRIDE (not a BASELINE config): '}{net}, {H}x{W} 1->1-ch, "
                    f"1000-step DDPM, batch {B} per GPU; a step = 1 of the 1000 denoising steps for the whole batch")
        out = {
            "metric": "256x256 cDDPM slices/sec (1000-step)" if headline else None,
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
            "data": "synthetic" if not stub else "STUB (no GPU work: launcher rehearsal)",
            "config": {"workload": workload, "slices_per_gpu": B, "image": [H, W],
                       "sampler": "guided-diffusion DDPM, v-param, 1000 steps", "params": leg.n_params,
                       "sharding": f"slices x{world} (no data-path collective)"},
            "rel_l2_vs_cpu": rel,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "rccl_ranks": rccl_ranks,
            "per_rank_ms_per_step": main_res["per_rank_ms_per_step"],
            "per_rank_device_bytes": dev_bytes,
            "timing": main_res.get("timing"),
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "kernels": kernels,
            "modes": modes,
            "samplers": samplers,
            "workloads": workloads,
            "extra": {"gpu": leg.gpu_name, "compute_units": leg.n_cu, "finite_output": finite,
                      "workspace_GiB": round(info["workspace_bytes"] / 2 ** 30, 2), "launches_per_step": info["launches"],
                      "executed_flops_per_step": info["flops"],
                      "survey_flops_per_step": F_LIVE_PER_SLICE_STEP * B,
                      "hip_graph": bool(args.graph) and not stub, "winograd": bool(args.winograd) and not stub,
                      "launched_by": "self (bench.py --gpus N)" if os.environ.get("DSD_BENCH_LAUNCHED_BY_PARENT") else
                                     ("torch.distributed.run" if world > 1 else "single process"),
                      "backend": backend if world > 1 else None,
                      "seconds_per_1000_step_batch": round(ms_per_step, 3),
                      "setup_s": round(leg.t_setup, 1), "weight_broadcast_ms": leg.bcast_ms,
                      "weight_broadcast_buckets": getattr(leg, "bcast_buckets", None)},
        }
        if not headline:
            out["valid_for_baseline"] = False
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
