"""Time of 64 + 64 cross-sections (cg_slices_device) on a grid after the C4 flow (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fem_amd, bench
from fem_amd import flows
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flow = flows.flow_c4()
torch.cuda.set_device(0)
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})
masks = bench.build_masks(fem_amd, flows, g, specs, names, n, n, 0, 2.0 * n, 2.0 * n, torch.device("cuda", 0))
for s in flow: flows.apply_step(g, s, masks)
idx = [(512 * q + 256) % n for q in range(64)]
for rep in range(3):
    g.sync(); g.timer_start(); g.slices_device("YZ", idx); a = g.timer_stop()
    g.timer_start(); g.slices_device("XZ", idx); b = g.timer_stop()
    print("YZ %.3f ms  XZ %.3f ms" % (a, b))
import zlib, numpy as np
for ax in ("YZ", "XZ"):
    out = g.slices(ax, idx[:5])
    print(ax, "crc", zlib.crc32(b"".join(np.ascontiguousarray(x).tobytes() for t in out for x in t)))
