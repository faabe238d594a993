import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fem_amd as fem
from oracle import OracleGrid
n = 96
g = fem.ColumnGrid(2.0 * n, 2.0 * n, 2.0, 2.0, 2)
o = OracleGrid(2.0 * n, 2.0 * n, 2.0, 2.0, lmax=2)
rng = np.random.RandomState(5)
m = (rng.rand(n * n) < 0.6).astype(np.uint8)
gmask = g.mask_from_raster(m)
steps = [("d", None, 3, 10, 0), ("d", gmask, 4, 5, 0), ("d", None, 5, 6, 1.0), ("d", gmask, 6, 4, 0.8), ("e", gmask, 255, 7, 1.0)]
steps += [("d", None if k % 2 else gmask, 3 + k % 5, 2 + k % 3, (k % 3) / 2) for k in range(22)]
for idx, (kind, mk, mat, t, S) in enumerate(steps):
    om = None if mk is None else m
    pre = o.arrays(); preL = o.Lmax
    if kind == "d":
        g.deposit_general(mk, mat, t, S); o.deposit_general(om, mat, t, S)
    else:
        g.etch_general(mk, mat, t, S); o.etch_general(om, mat, t, S)
    ok = g.checksum() == o.checksum()
    ln, gm, gz = g.download(); oln, omat, ozp = o.arrays()
    print(idx, kind, "masked" if mk else "all", mat, t, S, "Lmax", g.Lmax, o.Lmax, "maxlen", ln.max(), oln.max(), "OK" if ok else "DIFF")
    if not ok:
        L = g.Lmax
        bad = np.nonzero((ln != oln) | (gm.reshape(-1, L) != omat.reshape(-1, L)).any(1) | (gz.reshape(-1, 2 * L) != ozp.reshape(-1, 2 * L)).any(1))[0]
        print("bad columns", bad.size, bad[:10])
        for c in bad[:4]:
            print(" col", c, divmod(c, n), "mask", m[c], "pre len", pre[0][c], "pre top mat", pre[1].reshape(-1, preL)[c][max(0, pre[0][c] - 1)])
            print("   gpu len", ln[c], gm.reshape(-1, L)[c][:ln[c] + 1], gz.reshape(-1, L, 2)[c][:ln[c] + 1].tolist())
            print("   ora len", oln[c], omat.reshape(-1, L)[c][:oln[c] + 1], ozp.reshape(-1, L, 2)[c][:oln[c] + 1].tolist())
        break
