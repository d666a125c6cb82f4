"""The N>1 path on CPU: world_size-2 gloo processes exercise the slice sharding, the one-off packed weight broadcast and
the final gather (the only collectives of the path; nothing is exchanged inside the step loop)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from diffusion_models_dsdiff_amd import parallel
    dist.init_process_group("gloo")
    rank, ws = parallel.world()
    assert ws == 2
    # 1) weights: rank 0 holds the real values, others garbage -> identical everywhere after ONE packed broadcast
    g = torch.Generator().manual_seed(5)
    params = {"a.weight": torch.randn(7, 3, generator=g), "b.bias": torch.randn(11, generator=g), "c": torch.randn(2, 2, 2, generator=g)}
    want = {k: v.clone() for k, v in params.items()}
    if rank != 0:
        for v in params.values():
            v.fill_(float("nan"))
    parallel.broadcast_packed(params, 0)
    for k in want:
        assert torch.equal(params[k], want[k]), k
    # 2) slices: [r::R] shards, a per-slice function of the slice index only, gathered back in order (ragged: 7 slices on 2 ranks)
    for n in (7, 8, 1, 0):
        mine = parallel.shard_indices(n, rank, ws)
        local = torch.tensor([[float(i) * 10 + 1] for i in mine]).reshape(len(mine), 1)
        full = parallel.gather_slices(local, n, 0)
        if rank == 0:
            assert full.shape == (n, 1)
            assert torch.equal(full[:, 0], torch.arange(n).float() * 10 + 1)
        else:
            assert full is None
    # shards are disjoint and complete
    allidx = sorted(parallel.shard_indices(9, 0, 2) + parallel.shard_indices(9, 1, 2))
    assert allidx == list(range(9))
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_single_process_is_identity():
    import torch
    from diffusion_models_dsdiff_amd import parallel
    assert parallel.world() == (0, 1)
    assert parallel.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    x = torch.arange(6.).reshape(3, 2)
    assert parallel.gather_slices(x, 3) is x


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start the two ranks ITSELF (the driver's
    call form), rendezvous, take the max over ranks, gather, and print ONE line with n_gpus = 2.  Rehearsed on the CPU:
    gloo backend + the stubbed compute leg (rank r sleeps 2(r+1) ms per step, so the max-over-ranks is rank 1's)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DSD_BENCH_BACKEND="gloo", DSD_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    assert len(out["per_rank_ms_per_step"]) == 2
    # the timed region is bracketed by barriers, so every rank's clock covers the slowest rank (4 ms per step, rank 1)
    assert out["ms_per_step"] == max(out["per_rank_ms_per_step"]) and min(out["per_rank_ms_per_step"]) >= 4.0
    assert out["gather_ms"] is not None and out["extra"]["weight_broadcast_ms"] is not None
    assert out["extra"]["launched_by"].startswith("self")
    # whole-job value: slices of ALL ranks / time
    assert abs(out["value"] - 2 * 16 / (1000.0 * out["ms_per_step"] / 1e3)) < 1e-3 * out["value"]
    assert out["metric"] is None and out["valid_for_baseline"] is False and "STUB" in out["data"]


def test_world_size_8_gloo_ragged_volume_and_bucketed_broadcast(tmp_path):
    """8 ranks (the node the driver scales to) on the CPU: a ragged 13-slice volume sharded [r::8] (ranks 5..7 get ONE slice,
    nobody none; 5 slices leave ranks 5..7 EMPTY) and gathered back in order; the bucketed weight broadcast where only rank 0
    holds values and the others never initialise theirs (parallel.broadcast_params_bucketed)."""
    worker = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from diffusion_models_dsdiff_amd import parallel
        dist.init_process_group("gloo")
        rank, ws = parallel.world()
        assert ws == 8
        for n in (13, 5, 8, 64):
            mine = parallel.shard_indices(n, rank, ws)
            assert len(mine) in (n // ws, n // ws + 1)
            local = torch.tensor([[float(i) * 3 + 2] for i in mine]).reshape(len(mine), 1)
            full = parallel.gather_slices(local, n, 0)
            if rank == 0:
                assert torch.equal(full[:, 0], torch.arange(n).float() * 3 + 2), (n, full)
        named = [("w%%d" %% i, (17 + i, 5)) for i in range(9)] + [("big", (4000,)), ("tail", (3,))]
        src = {nm: torch.randn(sh, generator=torch.Generator().manual_seed(k)) for k, (nm, sh) in enumerate(named)}
        got = {}
        nb = parallel.broadcast_params_bucketed(named, (lambda nm: src[nm]) if rank == 0 else None,
                                                lambda nm, t: got.__setitem__(nm, t.clone()), 0, None, 256)
        assert nb == len(parallel.plan_buckets(named, 256)) and nb >= 5
        for nm, sh in named:
            assert got[nm].shape == tuple(sh) and torch.equal(got[nm], src[nm]), nm    # (src is seeded: every rank can check)
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """)
    script = tmp_path / "worker8.py"
    script.write_text(worker % ROOT)
    port = free_port()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_bench_self_launch_eight_ranks_gloo():
    """The driver's N = 8 call form rehearsed on the CPU (stubbed compute leg, gloo): 8 ranks rendezvous, the bucketed weight
    broadcast runs, every repeat takes the max over the 8 ranks, per-rank device bytes and the timing block are in the line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DSD_BENCH_BACKEND="gloo", DSD_BENCH_STUB="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and len(out["per_rank_ms_per_step"]) == 8
    assert len(out["per_rank_device_bytes"]) == 8 and out["extra"]["weight_broadcast_buckets"] == 3
    assert out["timing"]["repeats"] == 3 and len(out["timing"]["ms_per_step_repeats"]) == 3
    assert out["ms_per_step"] == sorted(out["timing"]["ms_per_step_repeats"])[1] and out["ms_per_step"] >= 16.0   # rank 7 sleeps 16 ms / step
    assert abs(out["value"] - 8 * 16 / (1000.0 * out["ms_per_step"] / 1e3)) < 1e-3 * out["value"]


def test_bench_rejects_gpus_world_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", DSD_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "one rank per GPU" in r.stderr
