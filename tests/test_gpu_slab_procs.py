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
