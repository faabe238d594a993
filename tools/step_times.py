"""Per-step HIP-event times of a flow on one GPU (diagnostic)."""
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
REPS = int(os.environ.get("REPS", "2"))      # COMPACT=1 prints every pass after the first
for rep in range(REPS):
    g.createNewGrid(); g.sync()
    rows = []
    for k, s in enumerate(flow):
        g.stats_enable(rep == 0 and not os.environ.get("NOSTATS"))
        g.timer_start()
        flows.apply_step(g, s, masks)
        ms = g.timer_stop()
        st = g.stats() if (rep == 0 and not os.environ.get("NOSTATS")) else None
        rows.append((k, s, ms, st))
    if rep == 0:
        stats = [r[3] for r in rows]
        if os.environ.get("NOSTATS"):
            print("total %.2f ms | " % sum(r[2] for r in rows) + " ".join("%.2f" % r[2] for r in rows))
    elif os.environ.get("COMPACT"):
        v = [r[2] for r in rows]
        print("total %.2f ms | vdep %.2f vetch %.2f sdep %.2f setch %.2f | " % (
            sum(v), sum(x for x, r in zip(v, rows) if r[1]["kind"] != "ETCH" and r[1]["S"] == 0), sum(x for x, r in zip(v, rows) if r[1]["kind"] == "ETCH" and r[1]["S"] == 0),
            sum(x for x, r in zip(v, rows) if r[1]["kind"] != "ETCH" and r[1]["S"] > 0), sum(x for x, r in zip(v, rows) if r[1]["kind"] == "ETCH" and r[1]["S"] > 0))
            + " ".join("%.2f" % x for x in v))
    else:
        for (k, s, ms, _), st in zip(rows, stats):
            by = flows.algorithmic_bytes(s["kind"], st)
            print("%2d %-6s mask=%-6s mat=%3d t=%5.1f S=%.1f  %7.3f ms  %6.2f GB  %7.1f GB/s  stats=%s" % (
                k, s["kind"], s["mask"], s["mat"], s["t"], s["S"], ms, by / 1e9, by / 1e6 / ms, st[1:7]))
