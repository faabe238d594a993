"""1-D slab decomposition of the column grid over the GPUs of one node.

Rank r owns rows i in [r*nx_local, (r+1)*nx_local) of an (nx_local*world) x ny domain.  Columns
are independent except for the lateral stencil of conformal deposition / isotropic etch, whose
radius R reaches into the neighbouring slabs.  For those steps each rank
  1. snapshots its top heights (+ etchable flags)          grid.stencil_begin
  2. swaps R boundary rows with its two neighbours          torch.distributed send/recv (RCCL)
  3. runs the dilation / erosion and the column update      grid.stencil_finish
Layer stacks never move between GPUs.  Whole-domain scalars (max height, checksum) are
all-reduced.  The reference is single-threaded (template/script_columngrid.js); this file has
no counterpart there.

`grid` is any object with stencil_begin / stencil_finish / halo access in the style of
fem_amd.ColumnGrid; the tests drive the same class with a CPU stand-in over the gloo backend.
"""
import math

import torch
import torch.distributed as dist

STENCIL_DEPOSIT, STENCIL_ETCH = 0, 1


class SlabGrid:
    def __init__(self, grid, rank=0, world=1, device=None, group=None, dx=2.0, dy=2.0, ny=None, halo_rows=32):
        self.grid, self.rank, self.world, self.group = grid, rank, world, group
        self.device = device if device is not None else torch.device("cpu")
        self.dx, self.dy = dx, dy
        self.ny = ny if ny is not None else grid.NY
        self.has_lower, self.has_upper = rank > 0, rank < world - 1
        self._halo = None
        self._rows = 0
        if world > 1:
            self._alloc_halo(halo_rows)

    # ---- halo buffers live in torch so that torch.distributed can move them
    def _alloc_halo(self, rows):
        nbytes = rows * self.ny * 3
        self._halo = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(4)]
        self._rows = rows
        self.grid.set_halo_buffers(*[t.data_ptr() for t in self._halo], nbytes)

    def _radius(self, t, S):
        """Rows the step will exchange: 0 on the vertical fast path (script_columngrid.js:445, 609)."""
        t = max(0.0, float(t)); S = max(0.0, min(1.0, float(S)))
        if t <= 0 or t * S < min(self.dx, self.dy):
            return 0
        return int(math.ceil(t / self.dx))

    def _exchange(self, R):
        n = R * self.ny * 3
        s_lo, s_up, r_lo, r_up = [t[:n] for t in self._halo]
        ops = []
        if self.has_lower:
            ops += [dist.P2POp(dist.isend, s_lo, self.rank - 1, self.group), dist.P2POp(dist.irecv, r_lo, self.rank - 1, self.group)]
        if self.has_upper:
            ops += [dist.P2POp(dist.isend, s_up, self.rank + 1, self.group), dist.P2POp(dist.irecv, r_up, self.rank + 1, self.group)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def _step(self, kind, mask, mat, t, S):
        if self.world > 1:
            need = self._radius(t, S)
            if need > self._rows:
                self._alloc_halo(max(need, 2 * self._rows))
        R = self.grid.stencil_begin(kind, mask, mat, t, S, self.has_lower, self.has_upper)
        if R == 0:
            return
        if self.world > 1:
            self._exchange(R)
        self.grid.stencil_finish()

    # ---- the reference's step names
    def deposit_general(self, mask, matId, thickness, conformality=1.0):
        self._step(STENCIL_DEPOSIT, mask, matId, thickness, conformality)

    def etch_general(self, mask, matId, thickness, conformality=1.0):
        self._step(STENCIL_ETCH, mask, matId, thickness, conformality)

    def createNewGrid(self):
        self.grid.createNewGrid()

    # ---- whole-domain scalars
    def maxHeight(self):
        z = self.grid.maxHeight()
        if self.world > 1:
            t = torch.tensor([z], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            z = float(t.item())
        return z

    def checksum(self):
        """Sum of the per-slab checksums mod 2^64 = checksum of the undivided grid."""
        s = self.grid.checksum()
        if self.world > 1:
            parts = torch.tensor([s & 0xFFFFFFFF, s >> 32], dtype=torch.int64, device=self.device)
            dist.all_reduce(parts, op=dist.ReduceOp.SUM, group=self.group)
            lo, hi = int(parts[0].item()), int(parts[1].item())
            s = (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF
        return s
