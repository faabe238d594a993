"""Per-step times of a long flow without grid resets (diagnostic): every depositing launch takes a
plane, so the columns are compacted now and then; the spikes show what that costs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fem_amd
from fem_amd import flows
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
flow = flows.flow_c4()[1:]                       # without the substrate: the stack keeps its history
torch.cuda.set_device(0)
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})
masks = bench.build_masks(fem_amd, flows, g, specs, names, n, n, 0, 2.0 * n, 2.0 * n, torch.device("cuda", 0))
g.deposit_general(None, 3, 30, 0); g.sync()
for rep in range(reps):
    times = []
    for s in flow:
        g.timer_start(); flows.apply_step(g, s, masks); times.append(g.timer_stop())
    print("pass %d  total %7.2f ms  max step %6.2f ms  | %s" % (rep, sum(times), max(times), " ".join("%.2f" % t for t in times)), flush=True)
print("Lmax", g.Lmax, "max len", g.cavity_state()[2])
