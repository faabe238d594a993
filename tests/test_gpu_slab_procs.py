"""Slab driver with one PROCESS per rank on the GPU: two ranks share cuda:0 and swap halo and
ghost rows through torch.distributed (gloo moves device tensors; RCCL wants one GPU per rank)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_two_processes_one_gpu():
    env = dict(os.environ, PROBE_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29647", os.path.join(HERE, "dist_gpu_slab_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert all(f"slab flow {n} ok on 2 ranks" in r.stdout for n in ("c4", "cs", "cwet", "cw"))


@pytest.mark.gpu
def test_bench_strong_scaling_rehearsal():
    """bench.py --scaling strong with 2 ranks on one GPU (device tensors staged through gloo): the domain is the
    one of the 1-rank run cut into two slabs, so the checksums must agree; the bench's own check of the timed
    state runs in both."""
    import json
    root = os.path.dirname(HERE)
    common = ["--size", "256", "--flow", "c4", "--steps", "20", "--warmup", "3", "--no-cpu", "--repeats", "1"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=240)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    env = dict(os.environ, BENCH_ONE_GPU="1", BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", "strong"] + common
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    b = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["scaling"] == "strong" and b["n_gpus"] == 2 and b["config"]["world"] == 2
    assert b["config"]["grid"] == a["config"]["grid"] == [256, 256] and b["config"]["rows_per_gpu"] == 128
    assert b["checksum"] == a["checksum"]
    assert b["halo"]["bytes_sent_per_rank"] > 0 and a["halo"]["bytes_sent_per_rank"] == 0


@pytest.mark.gpu
def test_bench_weak_scaling_rehearsal():
    """bench.py with 2 ranks on one GPU in its default (weak) mode: every rank owns a slab of the N = 1 size that
    carries its own copy of the non-periodic masks, so the checksum of the two-slab domain is the sum of two
    one-slab grids whose patterns differ only by the phase of the 48 nm discs; the bench's own check of the timed
    state (fused launches against the statistics pass and an unfused-order replay) runs on both ranks."""
    import json
    root = os.path.dirname(HERE)
    common = ["--size", "256", "--flow", "c4", "--steps", "20", "--warmup", "20", "--no-cpu", "--repeats", "1"]
    env = dict(os.environ, BENCH_ONE_GPU="1", BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29657", os.path.join(root, "bench.py"), "--gpus", "2"] + common
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    b = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["scaling"] == "weak" and b["n_gpus"] == 2 and b["config"]["world"] == 2
    assert b["config"]["grid"] == [512, 256] and b["config"]["rows_per_gpu"] == 256
    assert b["halo"]["bytes_sent_per_rank"] > 0
    # a slab's lateral deposits take the vertical steps behind them along (three of them in a pass of C4), the run on the
    # reset grid goes out on its own; and the library's launch counters agree with the bench's rule for every kernel group
    assert b["fusion"]["fused_launches"] == 1 and b["fusion"]["lateral_deposits_with_steps_inside"] == 3
    assert b["schedule"] == "fold" and not [k for k, v in b["kernels"].items() if "launches_by_bench_rule" in v]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher around it): the parent starts the two ranks as a child process,
    passes rank 0's line through and leaves with the child's exit code; a rank count that does not match --gpus is
    refused instead of being reported."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BENCH_ONE_GPU="1", BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "256", "--steps", "20", "--warmup", "5",
           "--no-cpu", "--repeats", "3"]
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    b = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["config"]["world"] == 2 and b["steps"] == 20
    assert b["halo"]["bytes_sent_per_rank"] > 0
    assert b["timing"]["regions"] == 3 and b["timing"]["ms_per_step_min"] <= b["ms_per_step"] <= b["timing"]["ms_per_step_max"]
    # one rank started by a launcher while --gpus says two: refused
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "256", "--no-cpu"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)
