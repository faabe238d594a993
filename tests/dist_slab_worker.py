"""Worker of the multi-rank slab test: every rank owns one slab (CPU oracle engine), the ranks
exchange halos over torch.distributed (gloo) through fem_amd.slabs.SlabGrid, and the all-reduced
checksum must equal the checksum of the undivided grid after every step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from fem_amd import flows, reduce_mask                      # noqa: E402
from fem_amd.slabs import SlabGrid                          # noqa: E402
from oracle import OracleGrid                               # noqa: E402
from oracle_engine import OracleSlabEngine                  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nxl, ny = 40, 72
    nxg = nxl * world
    flow = flows.FLOWS[os.environ.get("SLAB_FLOW", "c4")]()
    specs = flows.mask_specs(2.0 * nxg)
    whole = OracleGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    full_masks, my_masks = {}, {}
    for name in sorted({s["mask"] for s in flow if s["mask"]}):
        red = reduce_mask(flows.shapes_of(specs[name], 2 * nxg, 2 * ny))
        r = whole.mask_raster(red)
        full_masks[name] = r
        my_masks[name] = np.ascontiguousarray(r.reshape(nxg, ny)[rank * nxl:(rank + 1) * nxl]).reshape(-1)
    eng = OracleSlabEngine(nxg, ny, rank * nxl, nxl)
    slab = SlabGrid(eng, rank, world, device=torch.device("cpu"), ny=ny, halo_rows=4)     # small: forces a re-allocation at R=15
    for k, s in enumerate(flow):
        flows.apply_step(slab, s, my_masks)
        flows.apply_step(whole, s, full_masks)
        got, want = slab.checksum(), whole.checksum()
        assert got == want, f"rank {rank} step {k} {s}: {got:016x} != {want:016x}"
        assert slab.maxHeight() == whole.max_height()
    if rank == 0:
        print("slab flow ok on", world, "ranks")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
