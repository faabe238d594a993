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
lines = np.linspace(0, n - 1, 64).astype(np.int32)
def layers():
    """layers in the grid, from 64 lines of it"""
    return float(g.slices("YZ", lines)[0].mean()) * n * n
model = {"k_wet_step": [0, 0.0], "k_ald_coat": [0, 0.0]}       # launches, algorithmic bytes (fem_amd/flows.py)
def timed(label, fn, kernel, launches):
    lb = layers()
    g.sync(); t0 = time.time(); fn(); g.sync(); dt = time.time() - t0
    la = layers()
    # (the stacks of every launch of a call are taken as the mean of the call's first and last state)
    if kernel == "k_wet_step":
        b = flows.wet_step_bytes(n * n, (lb + la) / 2, (lb + la) / 2)
    else:
        b = flows.ald_coat_bytes(n * n, (lb + la) / 2, (lb + la) / 2, g.slider_sizes()[2])
    model[kernel][0] += launches; model[kernel][1] += launches * b
    print("%-34s %8.2f ms  %7.2f Gcol/s  Lmax %d  chk %016x  layers/column %.2f -> %.2f" % (label, dt * 1e3, n * n / dt / 1e9, g.Lmax, g.checksum(), lb / n / n, la / n / n))
timed("etch_wet(Ox, 4)  [2 increments]", lambda: g.etch_wet(4, 4), "k_wet_step", 2)
timed("deposit_ALD(6, 4) [1 incr + final]", lambda: g.deposit_ALD(6, 4), "k_ald_coat", 2)
timed("etch_wet(ALL, 2)  [1 increment]", lambda: g.etch_wet(255, 2), "k_wet_step", 1)
timed("deposit_ALD(7, 6) [2 incr + final]", lambda: g.deposit_ALD(7, 6), "k_ald_coat", 3)
for k, (cnt, b) in model.items():
    print("MODEL %s launches %d algorithmic_bytes %d" % (k, cnt, int(b)))
