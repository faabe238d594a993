"""1-D slab decomposition of the column grid over the GPUs of one node.

Rank r owns rows i in [r*nx_local, (r+1)*nx_local) of an (nx_local*world) x ny domain.  Columns
are independent except for the lateral stencil of conformal deposition / isotropic etch, whose
radius R reaches into the neighbouring slabs.  For those steps each rank
  1. packs the top heights (+ etchable flags) of its R boundary rows   grid.stencil_begin
  2. swaps R boundary rows with its two neighbours          torch.distributed send/recv (RCCL)
     while the rows that reach no neighbour are updated     grid.stencil_interior
  3. runs the dilation / erosion and the column update      grid.stencil_finish (the rows left)
Layer stacks never move between GPUs.  Whole-domain scalars (max height, checksum, CMP's
target height, "is there any gap") are all-reduced.  strip_connected / identify_cavity flood
across slab edges: each rank relaxes to its local fixpoint, swaps one boundary row (whole
columns with their flood marks) with each neighbour, and the ranks stop when a round changed
nothing anywhere.  etch_wet swaps the same ghost row before every increment (a cut reaches one
column sideways); deposit_ALD swaps it once, then the seeds of the rows a coat can reach, once.
The reference is single-threaded (template/script_columngrid.js); this file has
no counterpart there.

`grid` is any object with stencil_begin / stencil_finish / halo access in the style of
fem_amd.ColumnGrid; the tests drive the same class with a CPU stand-in over the gloo backend.
"""
import contextlib
import math
import time

import torch
import torch.distributed as dist

STENCIL_DEPOSIT, STENCIL_ETCH = 0, 1


class TorchComm:
    """Neighbour swaps and small all-reduces over torch.distributed (backend nccl = RCCL, or gloo).

    RCCL moves device buffers directly.  gloo is a host-memory transport: device buffers are staged
    through host tensors here (its own handling of device tensors hung now and then in an all-reduce
    with two ranks on one GPU), which is also what the CPU tests exercise."""

    def __init__(self, rank, world, device, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.stage = dist.get_backend(group) == "gloo" and torch.device(device).type != "cpu"

    def start_swap(self, send_lower, send_upper, recv_lower, recv_upper):
        """send_lower -> rank-1 (arrives as its recv_upper), send_upper -> rank+1 (its recv_lower).
        Returns what finish_swap waits for; with nccl the transfers run on RCCL's own stream, after
        the work already queued on the current one."""
        back = []
        if self.stage:
            torch.cuda.current_stream().synchronize()
            host = [send_lower.cpu(), send_upper.cpu(), torch.empty_like(recv_lower, device="cpu"), torch.empty_like(recv_upper, device="cpu")]
            back = [(recv_lower, host[2], self.rank > 0), (recv_upper, host[3], self.rank < self.world - 1)]
            send_lower, send_upper, recv_lower, recv_upper = host
        ops = []
        if self.rank > 0:
            ops += [dist.P2POp(dist.isend, send_lower, self.rank - 1, self.group), dist.P2POp(dist.irecv, recv_lower, self.rank - 1, self.group)]
        if self.rank < self.world - 1:
            ops += [dist.P2POp(dist.isend, send_upper, self.rank + 1, self.group), dist.P2POp(dist.irecv, recv_upper, self.rank + 1, self.group)]
        return (dist.batch_isend_irecv(ops) if ops else []), back

    def finish_swap(self, pending):
        reqs, back = pending
        for req in reqs:
            req.wait()
        for dev_t, host_t, used in back:
            if used:
                dev_t.copy_(host_t)

    def swap(self, send_lower, send_upper, recv_lower, recv_upper):
        self.finish_swap(self.start_swap(send_lower, send_upper, recv_lower, recv_upper))

    def reduce(self, values, op="max", dtype=torch.int64):
        t = torch.tensor(list(values), dtype=dtype, device="cpu" if self.stage else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
        return t.tolist()


class SlabGrid:
    """stream: the torch.cuda.Stream everything of this slab is ordered on when the communicator moves
    device buffers itself (RCCL).  The handle's kernels and the communication calls must share ONE
    stream: RCCL orders its transfers against torch's current stream only, and a handle's own stream
    is a non-blocking one that nothing else waits for.  Given None, the driver makes a stream of its
    own; either way it points the handle at it (grid.set_stream) and issues every communication call
    with it current."""

    def __init__(self, grid, rank=0, world=1, device=None, group=None, dx=2.0, dy=2.0, ny=None, halo_rows=32, comm=None,
                 stream=None):
        self.grid, self.rank, self.world, self.group = grid, rank, world, group
        self.device = device if device is not None else torch.device("cpu")
        self.comm = comm if comm is not None else (TorchComm(rank, world, self.device, group) if world > 1 else None)
        self.dx, self.dy = dx, dy
        self.ny = ny if ny is not None else grid.NY
        self.has_lower, self.has_upper = rank > 0, rank < world - 1
        self._halo = None
        self._rows = 0
        self._gaps_anywhere = None      # "some slab may hold a gap or a cavity": None = ask the ranks again
        self.halo_bytes = 0             # bytes this rank has sent in halo exchanges of lateral steps
        self.swap_seconds = 0.0         # host time between the start and the completion of those exchanges
        self.stream = None
        if self.device_side_comm():
            self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
            self.grid.set_stream(self.stream.cuda_stream)
        if world > 1:
            self._alloc_halo(halo_rows)

    def device_side_comm(self):
        """True when the communicator hands device buffers to the transport as they are (RCCL), i.e. when
        ordering between the handle's kernels and the transfers is a matter of streams."""
        return (self.comm is not None and not getattr(self.comm, "stage", False)
                and torch.device(self.device).type == "cuda")

    def _on_stream(self):
        """Context in which communication calls and torch allocations are issued: the slab's stream current."""
        if isinstance(self.stream, torch.cuda.Stream):
            return torch.cuda.stream(self.stream)
        return contextlib.nullcontext()

    def _zeros(self, nbytes, count=4):
        """Device (or host) byte buffers, cleared and ready for the handle's stream."""
        with self._on_stream():
            bufs = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(count)]
        if self.stream is None and torch.device(self.device).type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()      # the handle's own stream does not wait for torch's
        return bufs

    # ---- halo buffers live in torch so that torch.distributed can move them
    def _alloc_halo(self, rows):
        nbytes = rows * self.ny * 3
        self._halo = self._zeros(nbytes)
        self._rows = rows
        self.grid.set_halo_buffers(*[t.data_ptr() for t in self._halo], nbytes)

    def _radius(self, t, S):
        """Rows the step will exchange: 0 on the vertical fast path (script_columngrid.js:445, 609)."""
        t = max(0.0, float(t)); S = max(0.0, min(1.0, float(S)))
        if t <= 0 or t * S < min(self.dx, self.dy):
            return 0
        return int(math.ceil(t / self.dx))

    def _exchange(self, R):
        """Swap the R halo rows; the rows that need none of them are computed while they travel."""
        n = R * self.ny * 3
        bufs = [t[:n] for t in self._halo]
        interior = getattr(self.grid, "stencil_interior", None)
        if not self.device_side_comm():
            self.grid.sync()                     # the rows are read by the host / by torch, not on the grid's stream
        t0 = time.perf_counter()
        with self._on_stream():
            if interior is None or not hasattr(self.comm, "start_swap"):
                self.comm.swap(*bufs)
            else:
                reqs = self.comm.start_swap(*bufs)
                interior()
                self.comm.finish_swap(reqs)
        self.swap_seconds += time.perf_counter() - t0
        self.halo_bytes += n * (int(self.has_lower) + int(self.has_upper))

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

    def _reduce_max(self, values):
        with self._on_stream():
            return [int(v) for v in self.comm.reduce([int(v) for v in values], "max")]

    def _swap_ghosts(self, nbytes):
        """My row 0 becomes the lower neighbour's upper ghost, my last row the upper neighbour's lower ghost."""
        with self._on_stream():
            self.comm.swap(*[t[:nbytes] for t in self._ghost])

    def _open_somewhere(self, mask):
        """Does the mask open a cell anywhere in the domain?  One all-reduce per mask OBJECT (rasters never change):
        the answer is kept on the object itself, not under its id -- the library hands the id of a destroyed mask
        out again, and the next raster in that slot must not inherit the answer."""
        if mask is None or not hasattr(self.grid, "mask_any_open"):
            return True
        try:
            return mask._open_somewhere
        except AttributeError:
            pass
        ans = bool(self._reduce_max([1 if self.grid.mask_any_open(mask) else 0])[0])
        try:
            mask._open_somewhere = ans
        except AttributeError:            # a mask type that takes no attributes (a numpy raster of the CPU stand-in): ask every time
            pass
        return ans

    # ---- the reference's step names
    def deposit_general(self, mask, matId, thickness, conformality=1.0):
        lateral = self._radius(thickness, conformality) > 0
        if self.world > 1 and lateral and not self._open_somewhere(mask):
            # no open cell in the whole domain: script_columngrid.js:485 returns -- after line 470 has put the
            # step's kernel into the kernel cache, which a later step with a colliding key will find
            touch = getattr(self.grid, "kernel_cache_touch", None)
            if touch is not None and float(thickness) > 0:
                touch(thickness, conformality)
            return
        if matId == 2:
            self._gaps_anywhere = None                   # material 2 is what identify_cavity strips
        self._step(STENCIL_DEPOSIT, mask, matId, thickness, conformality)
        if self.world > 1 and lateral:
            self.identify_cavity()                       # script_columngrid.js:543; a slab handle leaves it to us

    def invalidate(self):
        """Call after changing the handle behind this driver's back (upload, snapshot load ...)."""
        self._gaps_anywhere = None

    def strip_connected(self, matId):
        """strip_connected(matId) (script_columngrid.js:867 - script_columngrid2.js:1) over all slabs."""
        if self.world == 1:
            return self.grid.strip_connected(matId)
        g = self.grid
        self._gaps_anywhere = None
        planes = self._reduce_max([g.cavity_state()[2]])[0]
        nbytes, s_lo, s_up, lo, up = self._ghost_buffers(planes)
        g.ghost_pack(planes, s_lo, s_up); g.sync(); self._swap_ghosts(nbytes)
        g.strip_seed(matId, planes, lo, up)
        g.ghost_pack(planes, s_lo, s_up); g.sync(); self._swap_ghosts(nbytes)
        while True:
            changed = g.strip_sweep(matId, planes, lo, up)
            if not self._reduce_max([changed])[0]:
                break
            g.ghost_pack(planes, s_lo, s_up); g.sync(); self._swap_ghosts(nbytes)
        g.strip_finish()

    def _ghost_buffers(self, planes):
        """(bytes per message, send lower, send upper, received lower or 0, received upper or 0)."""
        nbytes = self.grid.ghost_bytes(planes)
        if getattr(self, "_ghost", None) is None or self._ghost[0].numel() < nbytes:
            self._ghost = self._zeros(nbytes)
        s_lo, s_up, r_lo, r_up = [t.data_ptr() for t in self._ghost]
        return nbytes, s_lo, s_up, (r_lo if self.has_lower else 0), (r_up if self.has_upper else 0)

    def etch_wet(self, matId, totalThk, dt=None):
        """etch_wet(matId, totalThk, {dt}) (script_columngrid2.js:893 - script_columngrid3.js:308) over all
        slabs, without the slider cache: per increment one ghost row per side is swapped (a cut reaches
        one column sideways), the increment runs on every slab against the whole domain's height, and
        the cavity pass follows over all slabs."""
        if self.world == 1:
            return self.grid.etch_wet(matId, totalThk, dt=dt)
        g = self.grid
        T = 0.0 if totalThk != totalThk else max(0.0, float(totalThk))                       # cg2:894-895
        if T <= 0:
            return
        base_dt = min(self.dx, self.dy) if dt is None else float(dt)                          # cg2:905
        if not base_dt > 1e-6:
            base_dt = 1e-6
        zs = g.zScale
        R, t = 0.0, 0.0
        while t < T - 1e-9:                                                                   # cg3:286-306
            R += 0.7071067811865476
            diag = False
            if R > 1:
                R -= 1; diag = True
            dt_q = max(1, int(g._js_round(min(base_dt, T - t) / zs)))
            _, _, max_len, top_q = g.cavity_state()
            planes, hmax_q = self._reduce_max([max_len, top_q])
            nbytes, s_lo, s_up, lo, up = self._ghost_buffers(planes)
            g.ghost_pack(planes, s_lo, s_up); g.sync(); self._swap_ghosts(nbytes)
            g.etch_wet_step_slab(matId, dt_q, diag, hmax_q, planes, lo, up)
            self._gaps_anywhere = None
            self.identify_cavity()                                                            # cg3:267
            t += dt_q * zs
        self.identify_cavity()                                                                # cg3:307
        g.release_scratch()

    def _seed_buffers(self, nbytes):
        if getattr(self, "_seedbuf", None) is None or self._seedbuf[0].numel() < nbytes:
            self._seedbuf = self._zeros(max(nbytes, 8))
        s_lo, s_up, r_lo, r_up = [t.data_ptr() for t in self._seedbuf]
        return s_lo, s_up, (r_lo if self.has_lower else 0), (r_up if self.has_upper else 0)

    def deposit_ALD(self, matId, totalThk, dt=None):
        """deposit_ALD(matId, totalThk, {dt}) (script_columngrid2.js:111-868) over all slabs, without the
        slider cache.  The seeds are collected once from the grid as it stands (cg2:826), so the slabs
        exchange once: a ghost row per side for the side faces of the boundary rows, then the seeds of
        the ceil(totalThk/dx) rows next to each edge, which every later coat reads.  Each increment
        then needs the whole domain's height and the cavity pass over all slabs, nothing else."""
        if self.world == 1:
            return self.grid.deposit_ALD(matId, totalThk, dt=dt)
        g = self.grid
        T = 0.0 if totalThk != totalThk else max(0.0, float(totalThk))                       # cg2:112-113
        if T <= 0:
            return
        step = max(max(self.dx, self.dy, min(self.dx, self.dy)) if dt is None else float(dt), 1e-3)   # cg2:124-127
        _, _, max_len, top_q = g.cavity_state()
        planes, hmax_q = self._reduce_max([max_len, top_q])
        nbytes, s_lo, s_up, lo, up = self._ghost_buffers(planes)
        g.ghost_pack(planes, s_lo, s_up); g.sync(); self._swap_ghosts(nbytes)
        seed_planes = self._reduce_max([g.ald_collect_seeds_slab(T, hmax_q, planes, lo, up)])[0]
        nbytes = g.ald_seed_halo_bytes(seed_planes)
        s_lo, s_up, lo, up = self._seed_buffers(nbytes)
        g.ald_seed_halo_pack(seed_planes, s_lo, s_up); g.sync()
        with self._on_stream():
            self.comm.swap(*[t[:nbytes] for t in self._seedbuf])
        g.ald_seed_halo_unpack(seed_planes, lo, up); g.sync()
        try:
            t = step
            while t < T - 1e-9:                                                               # cg2:843-858
                hbase_q = self._reduce_max([g.cavity_state()[3]])[0]
                g.ald_coat_slab(matId, t, hbase_q)
                self._gaps_anywhere = None
                self.identify_cavity()                                                        # cg2:851
                g.ald_merge_slab(hbase_q)
                self._gaps_anywhere = None
                t += step
            g.ald_coat_slab(matId, T, self._reduce_max([g.cavity_state()[3]])[0])             # cg2:861-862
            self._gaps_anywhere = None
            self.identify_cavity()                                                            # cg2:867
        finally:
            g.release_scratch()
            g.ald_drop_seeds()

    def identify_cavity(self):
        """identify_cavity() (script_columngrid2.js:9-83) over all slabs."""
        if self.world == 1:
            return self.grid.identify_cavity()
        # the flags only change in strip / CMP / cavity passes (all of them collective calls of this
        # class), so the answer of the last all-reduce stands until one of those has run
        if self._gaps_anywhere is None:
            gaps, cav, _, _ = self.grid.cavity_state(flags_only=True)
            self._gaps_anywhere = bool(self._reduce_max([1 if (gaps or cav) else 0])[0])
        if not self._gaps_anywhere:
            return
        top_q = self._reduce_max([self.grid.cavity_state()[3]])[0]
        zs = self.grid.zScale
        hmax_q = top_q if top_q > 0 else int(math.floor(1.0 / zs + 0.5)) & 0xFFFF      # maxHeight() is 1 on an empty grid
        self.grid.cavity_fill(hmax_q)
        self.strip_connected(2)
        self._gaps_anywhere = None

    def cmp(self, depth, stopMatId=0):
        """cmp(depth, stopMatId) (script_columngrid.js:747-859) over all slabs."""
        if self.world == 1:
            return self.grid.cmp(depth, stopMatId)
        if not depth > 0:
            return
        top_q, stop_q, found = self._reduce_max(self.grid.cmp_scan(stopMatId))
        zs = self.grid.zScale
        z_top = top_q * zs
        target = z_top - depth if (stopMatId == 0 or not found) else max(z_top - depth, stop_q * zs)
        self.grid.cmp_apply(target, stopMatId)
        self._gaps_anywhere = None
        self.identify_cavity()

    def etch_general(self, mask, matId, thickness, conformality=1.0):
        self._step(STENCIL_ETCH, mask, matId, thickness, conformality)

    def createNewGrid(self):
        self.grid.createNewGrid()
        self._gaps_anywhere = False      # an empty grid has neither

    # ---- whole-domain scalars
    def maxHeight(self):
        z = self.grid.maxHeight()
        if self.world > 1:
            with self._on_stream():
                z = float(self.comm.reduce([z], "max", torch.float64)[0])
        return z

    def checksum(self):
        """Sum of the per-slab checksums mod 2^64 = checksum of the undivided grid."""
        s = self.grid.checksum()
        if self.world > 1:
            with self._on_stream():
                lo, hi = [int(v) for v in self.comm.reduce([s & 0xFFFFFFFF, s >> 32], "sum")]
            s = (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF
        return s
