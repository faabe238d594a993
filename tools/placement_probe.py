"""Does the speed of the plane-writing steps depend on where a grid's planes land in memory?
A ballast allocation of BALLAST_GB made before the grid shifts everything the grid allocates (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fem_amd
from fem_amd import flows
import bench

n = 16384
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0
free_it = len(sys.argv) > 2 and sys.argv[2] == "free"
flow = flows.flow_c4()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ballast = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device=dev) if gb > 0 else None
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
if free_it:
    del ballast; torch.cuda.empty_cache()
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})
masks = bench.build_masks(fem_amd, flows, g, specs, names, n, n, 0, 2.0 * n, 2.0 * n, dev)
best = None
for rep in range(3):
    g.createNewGrid(); g.sync()
    v = []
    for s in flow:
        g.timer_start(); flows.apply_step(g, s, masks); v.append(g.timer_stop())
    if rep: best = v if best is None or sum(v) < sum(best) else best
v = best
print("ballast %5.1f GB%s  total %.2f ms  appends %.2f %.2f %.2f  pops %.2f %.2f  substrate %.2f" % (gb, " (freed)" if free_it else "", sum(v), v[1], v[2], v[9], v[3], v[8], v[0]), flush=True)
