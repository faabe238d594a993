"""Timing of etch_wet / deposit_ALD on a patterned grid (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fem_amd
from fem_amd import flows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = fem_amd.ColumnGrid(2 * n, 2 * n, 2, 2)
g.keep_step_snapshots = False
import torch
specs = flows.mask_specs(2 * n)
rasters = {k: flows.raster_torch(specs[k], g.NX, g.NY, 0, "cuda") for k in ("M", "Mc")}
torch.cuda.synchronize()
masks = {k: g.mask_from_device_raster(r.data_ptr(), r.numel()) for k, r in rasters.items()}
for mat, t in ((3, 30), (4, 8), (5, 12)):
    g.deposit_general(None, mat, t, 0)
g.etch_general(masks["M"], 255, 25, 0)
g.sync()
def timed(label, fn):
    g.sync(); t0 = time.time(); fn(); g.sync(); dt = time.time() - t0
    print("%-34s %8.2f ms  %7.2f Gcol/s  Lmax %d  chk %016x" % (label, dt * 1e3, n * n / dt / 1e9, g.Lmax, g.checksum()))
timed("etch_wet(Ox, 4)  [2 increments]", lambda: g.etch_wet(4, 4))
timed("deposit_ALD(6, 4) [1 incr + final]", lambda: g.deposit_ALD(6, 4))
timed("etch_wet(ALL, 2)  [1 increment]", lambda: g.etch_wet(255, 2))
timed("deposit_ALD(7, 6) [2 incr + final]", lambda: g.deposit_ALD(7, 6))
