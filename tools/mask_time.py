"""Times the shape rasteriser (cg_mask_from_shapes) on the benchmark's periodic masks at full size and
checks every raster against the closed form the benchmark uses.  python tools/mask_time.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fem_amd as fem                      # noqa: E402
from fem_amd import flows                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
specs = flows.mask_specs(2.0 * n)
g = fem.ColumnGrid(2.0 * n, 2.0 * n, 2, 2)
dev = torch.device("cuda", 0)
for name in ("M", "Mshift", "Mc", "Mr"):
    t0 = time.time()
    data = flows.shapes_of(specs[name], 2 * n, 2 * n)
    t0 = time.time()
    red = fem.reduce_mask(data, packed=True)
    t1 = time.time()
    m = g.mask_from_shapes(red)            # returns after the raster is complete
    t2 = time.time()
    line = f"{name}: {len(red['shapes'])} shapes, reduce_mask {t1 - t0:.2f} s, mask_from_shapes {1e3 * (t2 - t1):.1f} ms"
    if specs[name]["kind"] != "shapes":
        want = flows.raster_torch(specs[name], n, n, 0, dev)
        got = torch.from_numpy(m.download()).to(dev)
        line += ", equals the closed form" if bool((got == want).all()) else ", DIFFERS from the closed form"
    print(line, flush=True)
    m.destroy()
g.close()
