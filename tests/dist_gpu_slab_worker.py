"""Worker of test_gpu_slab_procs.py: one rank of a slab run whose ranks are separate processes
sharing GPU 0.  RCCL refuses two ranks on one device ("Duplicate GPU detected"), so on a one-GPU
box the exchange goes over gloo with device tensors (PROBE_BACKEND=gloo); with one GPU per rank
the same worker runs over nccl.  Launch: python -m torch.distributed.run --nproc-per-node 2 <this file>"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(45, exit=True)      # a hung rank says where it hangs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.distributed as dist
import fem_amd
from fem_amd import flows
from fem_amd.slabs import SlabGrid
from oracle import OracleGrid

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
backend = os.environ.get("PROBE_BACKEND", "nccl")
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(backend)
rank, world = dist.get_rank(), dist.get_world_size()
nxl, ny = 64, 96
nxg = nxl * world
for name in ("c4", "cs", "cwet", "cw"):
    flow = flows.FLOWS[name]()
    specs = flows.mask_specs(2.0 * nxg)
    o = OracleGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    g = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(nxg, ny, rank * nxl, nxl))
    names = sorted({s["mask"] for s in flow if s["mask"]})
    reds = {n: fem_amd.reduce_mask(flows.shapes_of(specs[n], 2 * nxg, 2 * ny)) for n in names}
    om = {n: o.mask_raster(reds[n]) for n in names}
    gm = {n: g.mask_from_shapes(reds[n]) for n in names}
    slab = SlabGrid(g, rank, world, device=dev, ny=ny, halo_rows=4)
    if slab.device_side_comm():      # RCCL: the handle's kernels and the transfers share one stream
        assert slab.stream is not None and g.get_stream() == slab.stream.cuda_stream
    for k, st in enumerate(flow):
        flows.apply_step(slab, st, gm)
        flows.apply_step(o, st, om)
        assert slab.checksum() == o.checksum(), (name, k, st)
    if rank == 0:
        print(backend, "slab flow", name, "ok on", world, "ranks", flush=True)
    g.close()
dist.destroy_process_group()
