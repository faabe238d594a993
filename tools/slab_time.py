"""What a slab costs next to an undivided grid of the same size, without any transport: the C4 flow on ONE slab handle
of n x n columns that has a neighbour on both sides (the middle slab of three), driven through SlabGrid with a
communicator that moves nothing -- the received rows are whatever the buffers hold, so the result means nothing, the
launches and their durations are those of a rank in a weak-scaling run.  python tools/slab_time.py [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fem_amd                             # noqa: E402
from fem_amd import flows                  # noqa: E402
from fem_amd.slabs import SlabGrid         # noqa: E402


class NoTransport:
    stage = False

    def swap(self, *bufs):
        pass

    def start_swap(self, *bufs):
        return []

    def finish_swap(self, reqs):
        pass

    def reduce(self, values, op):
        return list(values)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flow = flows.FLOWS["c4"]()
specs = flows.mask_specs(2.0 * n)
names = sorted({s["mask"] for s in flow if s["mask"]})


def run(g, masks, label, sync):
    best = None
    for rep in range(4):
        g.createNewGrid(); sync()
        t0 = time.perf_counter()
        for s in flow:
            flows.apply_step(g, s, masks)
        sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("%-44s %7.2f ms per pass of the flow  %6.1f Gcol/s" % (label, best * 1e3, n * n * len(flow) / best / 1e9), flush=True)


g = fem_amd.ColumnGrid(2.0 * n, 2.0 * n, 2, 2)
masks = {nm: g.mask_from_shapes(fem_amd.reduce_mask(flows.shapes_of(specs[nm], 2 * n, 2 * n), packed=True)) for nm in names}
run(g, masks, "undivided grid", g.sync)
g.close()

stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    grid = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(3 * n, n, n, n))
    grid.set_stream(stream.cuda_stream)
    # (as bench.py lays the pattern out for a rank of a weak-scaling run: the undivided grid's pattern on this slab's rows)
    import bench
    masks = bench.build_masks(fem_amd, flows, grid, specs, names, n, n, n, 6.0 * n, 2.0 * n, torch.device("cuda:0"), 2.0 * n)
    slab = SlabGrid(grid, 1, 3, device="cuda:0", dx=2.0, dy=2.0, ny=n, stream=stream, comm=NoTransport())
    run(slab, masks, "middle slab of three, nothing transported", grid.sync)
