"""What it costs to cut a grid into slabs that the library drives (cg_group_*): the C4 flow on an n x n grid as one handle
and as a group of 2 and 4 slabs on the SAME device (peer copies between slabs; on a multi-GPU node the same calls go
over RCCL).  Times per pass of the flow, checksums compared.  python tools/group_time.py [n]   (CG_NO_FOLD=1 for the
one-handle figure on the slabs' own schedule)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fem_amd                             # noqa: E402
from fem_amd import flows                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flow = flows.FLOWS["c4"]()
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})


def run(g, label):
    masks = {nm: g.mask_from_shapes(fem_amd.reduce_mask(flows.shapes_of(specs[nm], 2 * n, 2 * n), packed=True)) for nm in names}
    best = None
    for rep in range(3):
        g.createNewGrid(); g.sync()
        t0 = time.perf_counter()
        for s in flow:
            flows.apply_step(g, s, masks)
        g.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    chk = g.checksum()
    print("%-34s %7.2f ms per pass of the flow  %6.1f Gcol/s  checksum %016x" % (label, best * 1e3, n * n * len(flow) / best / 1e9, chk), flush=True)
    return chk


g = fem_amd.ColumnGrid(2.0 * n, 2.0 * n, 2, 2)
ref = run(g, "one handle")
g.close()
for k in (2, 4):
    gg = fem_amd.GroupGrid(2.0 * n, 2.0 * n, 2, 2, devices=(0,) * k)
    chk = run(gg, "group of %d slabs, transport %s" % (k, gg.transport()))
    print("    identical" if chk == ref else "    DIFFERS")
    gg.close()
