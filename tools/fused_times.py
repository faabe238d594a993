"""Times of the launches of a flow on one GPU with runs of vertical steps fused (diagnostic): every maximal run of
consecutive vertical steps is timed as one piece, every lateral step on its own.  CG_NO_FUSE=1 for the unfused times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fem_amd
from fem_amd import flows
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flow = flows.FLOWS[sys.argv[2] if len(sys.argv) > 2 else "c4"]()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})
masks = bench.build_masks(fem_amd, flows, g, specs, names, n, n, 0, 2.0 * n, 2.0 * n, dev)
vertical = lambda s: s["t"] * (0.0 if s["kind"] == "SUBSTR" else s["S"]) < 2.0
runs, cur = [], []
for k, s in enumerate(flow):
    if vertical(s):
        cur.append(k)
        if len(cur) == 6: runs.append(cur); cur = []
    else:
        if cur: runs.append(cur); cur = []
        runs.append([k])
if cur: runs.append(cur)
for rep in range(int(os.environ.get("REPS", "3"))):
    g.createNewGrid(); g.sync()
    out = []
    for r in runs:
        g.timer_start()
        for k in r: flows.apply_step(g, flow[k], masks)
        out.append((r, g.timer_stop()))
    print("total %.2f ms | " % sum(ms for _, ms in out) + "  ".join("%s:%.2f" % ("+".join(map(str, r)), ms) for r, ms in out))
print("checksum %016x" % g.checksum())
