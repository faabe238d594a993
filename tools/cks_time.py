import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, fem_amd
from fem_amd import flows
import bench
n = 16384
flow = flows.FLOWS["c4"]()
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})
t0 = time.time(); masks = bench.build_masks(fem_amd, flows, g, specs, names, n, n, 0, 2.0 * n, 2.0 * n, dev); torch.cuda.synchronize(); print("mask build %.1f ms" % ((time.time() - t0) * 1e3))
g.createNewGrid()
for s in flow: flows.apply_step(g, s, masks)
g.sync()
for k in range(3):
    g.timer_start(); c = g.checksum(); ms = g.timer_stop(); print("checksum %016x %.3f ms" % (c, ms))
