"""Isotropic etch that takes a whole film off (thickness above the film's): time of the lateral step with and
without the rows-without-neighbours phase of the streaming kernel (CG_STREAM_NOFAST=1), on one GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fem_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
for t_film, t_etch, S in ((10, 12, 1.0), (30, 33, 0.5), (10, 10, 1.0)):
    for rep in range(2):
        g.createNewGrid()
        g.deposit_general(None, 3, 30, 0); g.deposit_general(None, 5, t_film, 0); g.sync()
        g.timer_start(); g.etch_general(None, 5, t_etch, S); ms = g.timer_stop()
    print("film %g nm, isotropic etch %g nm S=%g: %.3f ms  checksum %016x" % (t_film, t_etch, S, ms, g.checksum()))
