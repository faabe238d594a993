"""Multi-rank slab decomposition on the CPU (gloo): P slabs with halo exchange == one grid."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,flow", [(2, "c4"), (3, "c4"), (2, "cs"), (3, "cs"), (2, "cwet"), (2, "cw"), (3, "cw")])
def test_slab_flow_matches_whole_grid(world, flow):
    """c4: halo rows of the lateral steps.  cs: CMP, STRIP and identify_cavity across slab edges
    (all-reduced scalars and the ghost-row flood of fem_amd/slabs.py).  cwet: etch_wet increments with
    one ghost row per side.  cw: deposit_ALD as well (one ghost row, then the seeds of the rows a coat reaches)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", SLAB_FLOW=flow)
    port = 29500 + world + (10 if flow == "cs" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_slab_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert f"slab flow ok on {world} ranks" in p.stdout
