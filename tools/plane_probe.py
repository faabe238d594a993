"""Time of one append per plane: 31 unmasked deposits of alternating materials on an empty grid, each
pushes the old tops into the next plane (diagnostic for placement effects)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fem_amd
n = 16384
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
for rep in range(2):
    g.createNewGrid(); g.sync()
    v = []
    for k in range(31):
        g.timer_start(); g.deposit_general(None, 3 + (k & 1), 2.0, 0.0); v.append(g.timer_stop())
    if rep:
        print("append into plane P (ms):", " ".join("%.2f" % x for x in v[1:]), flush=True)
