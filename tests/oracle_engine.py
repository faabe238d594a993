"""CPU stand-in for one rank's grid in the slab driver tests (test infrastructure only).

It offers the same stencil_begin / stencil_finish / set_halo_buffers surface as
fem_amd.ColumnGrid, implemented with the CPU oracle, and speaks the same wire formats: halo =
R*ny u16 top heights followed by R*ny u8 etchable flags per direction; ghost row = len[ny]
(padded to 4 bytes) followed by rec[planes][ny] u32 records z1q | mat << 16 | flags << 24
(flags: 1 gap, 2 layer, 4 flood mark), a gap being an interval of its own."""
import ctypes
import math

import numpy as np

from oracle import OracleGrid


def _view(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))


class OracleSlabEngine:
    def __init__(self, nx_global, ny, i0, nx_local, dx=2.0, dy=2.0):
        self.o = OracleGrid(dx=dx, dy=dy, slab=(nx_global, ny, i0, nx_local))
        self.nx, self.ny, self.dx, self.dy = nx_local, ny, dx, dy
        self._halo = None
        self._pend = None

    NX = property(lambda s: s.nx)
    NY = property(lambda s: s.ny)

    def set_halo_buffers(self, s_lo, s_up, r_lo, r_up, nbytes):
        self._halo = [_view(p, nbytes) for p in (s_lo, s_up, r_lo, r_up)]

    def createNewGrid(self):
        self.o.reset()

    def mask_any_open(self, mask):
        return True if mask is None else bool(np.asarray(mask).any())

    def _tops_q(self):
        ln, mat, zp = self.o.arrays()
        L = self.o.Lmax
        z1 = zp.reshape(-1, L, 2)[:, :, 1]
        idx = np.maximum(ln.astype(np.int64) - 1, 0)
        q = z1[np.arange(ln.size), idx]
        q[ln == 0] = 0
        return q.reshape(self.nx, self.ny).astype(np.uint16)

    def stencil_begin(self, kind, mask, mat, t, S, has_lower, has_upper):
        t = max(0.0, float(t)); S = max(0.0, min(1.0, float(S)))
        if t <= 0:
            return 0
        if t * S < min(self.dx, self.dy):
            (self.o.etch_general if kind == 1 else self.o.deposit_general)(mask, mat, t, S)
            return 0
        R = int(math.ceil(t / self.dx))
        n = R * self.ny
        if has_lower or has_upper:
            q = self._tops_q()
            E = self.o.etch_flags(mask, mat) if kind == 1 else np.zeros((self.nx, self.ny), np.uint8)
            for buf, rows in ((self._halo[0], slice(0, R)), (self._halo[1], slice(self.nx - R, self.nx))):
                buf[:2 * n] = q[rows].reshape(-1).view(np.uint8)
                buf[2 * n:3 * n] = E[rows].reshape(-1)
        self._pend = (kind, mask, mat, t, S, has_lower, has_upper, R)
        return R

    def stencil_interior(self):
        """The engine updates the rows that need no halo here; this stand-in does all rows in
        stencil_finish, but the driver's overlapped exchange (start / interior / finish) still runs."""
        assert self._pend is not None
        self.interior_calls = getattr(self, "interior_calls", 0) + 1

    def stencil_finish(self):
        kind, mask, mat, t, S, has_lower, has_upper, R = self._pend
        self._pend = None
        n = R * self.ny

        def decode(buf):
            q = buf[:2 * n].view(np.uint16).astype(np.float64)
            H = (q * 0.1).astype(np.float32)          # float32 of q*zScale, as the height map stores it
            return np.ascontiguousarray(H), np.ascontiguousarray(buf[2 * n:3 * n].copy())

        lH = lE = uH = uE = None
        if has_lower:
            lH, lE = decode(self._halo[2])
        if has_upper:
            uH, uE = decode(self._halo[3])
        if kind == 1:
            self.o.etch_general_halo(mask, mat, t, S, lH, uH, lE, uE)
        else:
            self.o.deposit_general_halo(mask, mat, t, S, lH, uH)

    # ---- cmp / identify_cavity / strip_connected in the pieces the slab driver calls -----------
    zScale = 0.1

    def sync(self):
        pass

    def _layers(self):
        ln, mat, zp = self.o.arrays()
        L = self.o.Lmax
        return ln.astype(np.int64), mat.reshape(-1, L), zp.reshape(-1, L, 2).astype(np.int64), L

    def cmp_scan(self, stop):
        ln, mat, zp, L = self._layers()
        top_q, stop_q, found = 0, 0, 0
        for c in range(ln.size):
            if ln[c]:
                top_q = max(top_q, int(zp[c, ln[c] - 1, 1]))
                if stop:
                    for k in range(ln[c]):
                        if mat[c, k] == stop:
                            found = 1; stop_q = max(stop_q, int(zp[c, k, 1]))
        return top_q, stop_q, found

    def cmp_apply(self, target_z, stop):
        self.o.L.og_cmp_apply.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
        self.o.L.og_cmp_apply(self.o.h, float(target_z), int(stop))

    def cavity_state(self, flags_only=False):
        ln, mat, zp, L = self._layers()
        gaps = cav = 0; top_q = 0
        for c in range(ln.size):
            prev = 0
            for k in range(ln[c]):
                if zp[c, k, 0] > prev:
                    gaps = 1
                if mat[c, k] == 2:
                    cav = 1
                prev = zp[c, k, 1]
            if ln[c]:
                top_q = max(top_q, int(zp[c, ln[c] - 1, 1]))
        return gaps, cav, int(ln.max()) * 2 + 1 if ln.size else 0, top_q      # gaps are intervals of their own on the wire

    def etch_wet_step_slab(self, matId, dt_q, diagonal, hmax_q, planes, ghost_lower, ghost_upper):
        """One etch_wet increment of this slab: the ghost rows are put below / above the slab's own rows
        in a scratch grid, the reference's increment runs on that (a cut reaches one column sideways),
        and the slab keeps its own rows of the result."""
        ln, mat, zp, L = self._layers()
        lo, up = self._ghost_columns(planes, ghost_lower), self._ghost_columns(planes, ghost_upper)
        rows = (1 if lo is not None else 0) + self.nx + (1 if up is not None else 0)
        Lx = max(L, max([len(c) for g_ in (lo, up) if g_ is not None for c in g_] + [1]))
        eln = np.zeros(rows * self.ny, np.uint8); emat = np.zeros((rows * self.ny, Lx), np.uint8)
        ezp = np.zeros((rows * self.ny, Lx, 2), np.uint16)
        r0 = 0
        def put_ghost(cols, row):
            for j, lay in enumerate(cols):
                c = row * self.ny + j; eln[c] = len(lay)
                for k, (z0, z1, m, _) in enumerate(lay):
                    emat[c, k] = m; ezp[c, k] = (z0, z1)
        if lo is not None:
            put_ghost(lo, 0); r0 = 1
        a, b = r0 * self.ny, (r0 + self.nx) * self.ny
        eln[a:b] = ln; emat[a:b, :L] = mat; ezp[a:b, :L] = zp
        if up is not None:
            put_ghost(up, r0 + self.nx)
        scratch = OracleGrid(dx=self.dx, dy=self.dy, slab=(rows, self.ny, 0, rows), lmax=Lx)
        scratch.upload(eln, emat.reshape(-1), ezp.reshape(-1), Lx)
        scratch.etch_wet_step_part(matId, dt_q, diagonal, max(int(hmax_q), 1))
        sl, sm, sz = scratch.arrays(); Ls = scratch.Lmax
        sm = sm.reshape(-1, Ls)[a:b]; sz = sz.reshape(-1, Ls, 2)[a:b]
        self.o.upload(sl[a:b], sm.reshape(-1), sz.reshape(-1), Ls)

    def release_scratch(self):
        pass

    # ---- deposit_ALD in the pieces the slab driver calls ---------------------------------------
    def _with_rows(self, below, above, lo=None, up=None):
        """A scratch grid = `below` rows, the slab's own rows, `above` rows; lo / up = ghost columns for
        the single row next to the slab (else the extra rows are empty).  -> (grid, first own column, end)."""
        ln, mat, zp, L = self._layers()
        rows = below + self.nx + above
        Lx = max(L, max([len(c) for g_ in (lo, up) if g_ is not None for c in g_] + [1]))
        eln = np.zeros(rows * self.ny, np.uint8); emat = np.zeros((rows * self.ny, Lx), np.uint8)
        ezp = np.zeros((rows * self.ny, Lx, 2), np.uint16)

        def put_ghost(cols, row):
            for j, lay in enumerate(cols):
                c = row * self.ny + j; eln[c] = len(lay)
                for k, (z0, z1, m, _) in enumerate(lay):
                    emat[c, k] = m; ezp[c, k] = (z0, z1)
        if lo is not None:
            put_ghost(lo, below - 1)
        a, b = below * self.ny, (below + self.nx) * self.ny
        eln[a:b] = ln; emat[a:b, :L] = mat; ezp[a:b, :L] = zp
        if up is not None:
            put_ghost(up, below + self.nx)
        scratch = OracleGrid(dx=self.dx, dy=self.dy, slab=(rows, self.ny, 0, rows), lmax=Lx)
        scratch.upload(eln, emat.reshape(-1), ezp.reshape(-1), Lx)
        return scratch, a, b

    def _keep_own(self, scratch, a, b):
        sl, sm, sz = scratch.arrays(); Ls = scratch.Lmax
        self.o.upload(sl[a:b], sm.reshape(-1, Ls)[a:b].reshape(-1), sz.reshape(-1, Ls, 2)[a:b].reshape(-1), Ls)

    @staticmethod
    def _height_nm(q):
        return q * 0.1 if q > 0 else 1.0                      # maxHeight() is 1 on an empty grid

    def ald_collect_seeds_slab(self, T, hmax_q, planes, ghost_lower, ghost_upper):
        lo, up = self._ghost_columns(planes, ghost_lower), self._ghost_columns(planes, ghost_upper)
        below = 1 if lo is not None else 0
        scratch, a, b = self._with_rows(below, 1 if up is not None else 0, lo, up)
        si, sj, za, zb = scratch.ald_seeds_part(float(T), self._height_nm(int(hmax_q)))
        si = si - below
        own = (si >= 0) & (si < self.nx)                      # the ghost rows' own seeds are their slabs' business
        self._seed_rows = int(math.ceil(T / self.dx))
        self._seed_sides = (lo is not None, up is not None)
        self._seeds = [np.stack([si[own], sj[own], za[own].astype(np.int64), zb[own].astype(np.int64)], 1).astype(np.int64)]
        per_col = np.bincount(si[own] * self.ny + sj[own], minlength=self.nx * self.ny)
        return max(1, int(per_col.max()) if per_col.size else 1)

    def _seed_halo_layout(self, seed_planes):
        n = self._seed_rows * self.ny
        cb = (2 * n + 7) & ~7
        return n, cb, cb + seed_planes * n * 8

    def ald_seed_halo_bytes(self, seed_planes):
        return self._seed_halo_layout(seed_planes)[2]

    def ald_seed_halo_pack(self, seed_planes, send_lower, send_upper):
        n, cb, nb = self._seed_halo_layout(seed_planes)
        R = self._seed_rows
        own = self._seeds[0]
        for ptr, first in ((send_lower, 0), (send_upper, self.nx - R)):
            buf = _view(ptr, nb); buf[:] = 0
            cnt = buf[:2 * n].view(np.uint16)
            sd = buf[cb:].view(np.uint32).reshape(seed_planes, n, 2)
            for i, j, za, zb in own[(own[:, 0] >= first) & (own[:, 0] < first + R)]:
                e = (i - first) * self.ny + j
                sd[cnt[e], e] = (za, zb); cnt[e] += 1

    def ald_seed_halo_unpack(self, seed_planes, recv_lower, recv_upper):
        n, cb, nb = self._seed_halo_layout(seed_planes)
        R = self._seed_rows
        assert (bool(recv_lower), bool(recv_upper)) == self._seed_sides
        del self._seeds[1:]
        for ptr, first in ((recv_lower, -R), (recv_upper, self.nx)):
            if not ptr:
                continue
            buf = _view(ptr, nb)
            cnt = buf[:2 * n].view(np.uint16)
            sd = buf[cb:].view(np.uint32).reshape(seed_planes, n, 2)
            got = [(first + e // self.ny, e % self.ny, int(sd[q, e, 0]), int(sd[q, e, 1])) for e in range(n) for q in range(int(cnt[e]))]
            if got:
                self._seeds.append(np.array(got, np.int64))

    def ald_coat_slab(self, matId, t, hmax_q):
        """coat(t) on a scratch grid with room for the neighbours' seed rows; the slab keeps its own rows."""
        R = self._seed_rows
        ln, mat, zp, L = self._layers()
        self._ald_base = (ln.astype(np.uint8), mat.copy(), zp.astype(np.uint16), L)
        scratch, a, b = self._with_rows(R, R)
        sd = np.concatenate(self._seeds, 0)
        scratch.ald_coat_part(matId, t, self._height_nm(int(hmax_q)), sd[:, 0] + R, sd[:, 1], sd[:, 2], sd[:, 3])
        self._keep_own(scratch, a, b)

    def ald_merge_slab(self, hbase_q):
        ln, mat, zp, L = self._ald_base
        Ln = self.o.Lmax
        if Ln > L:                                             # the cavity pass made the arrays wider: the base follows
            m2 = np.zeros((ln.size, Ln), np.uint8); z2 = np.zeros((ln.size, Ln, 2), np.uint16)
            m2[:, :L] = mat; z2[:, :L] = zp
            mat, zp, L = m2, z2, Ln
        self.o.ald_merge_part(self._height_nm(int(hbase_q)), ln, mat.reshape(-1), zp.reshape(-1), L)

    def ald_drop_seeds(self):
        self._seeds = None; self._ald_base = None

    _js_round = staticmethod(lambda x: math.floor(x + 0.5))
    zScale = 0.1

    def cavity_fill(self, hmax_q):
        self.o.L.og_cavity_fill.argtypes = [ctypes.c_void_p, ctypes.c_double]
        self.o.L.og_cavity_fill(self.o.h, float(hmax_q) * 0.1)

    def ghost_bytes(self, planes):
        return ((self.ny + 3) & ~3) + planes * self.ny * 4

    def _intervals(self, c, ln, mat, zp, marks):
        """Column c as the wire's intervals: [(z1q, mat, flags)], gaps included."""
        out, prev = [], 0
        for k in range(ln[c]):
            if zp[c, k, 0] > prev:
                out.append((int(zp[c, k, 0]), 0, 1))
            fl = 2 | (4 if marks is not None and marks[c, k] else 0)
            out.append((int(zp[c, k, 1]), int(mat[c, k]), fl))
            prev = int(zp[c, k, 1])
        return out

    def ghost_pack(self, planes, send_lower, send_upper):
        ln, mat, zp, L = self._layers()
        marks = getattr(self, "_marks", None)
        nb = self.ghost_bytes(planes); lenb = (self.ny + 3) & ~3
        for ptr, row in ((send_lower, 0), (send_upper, self.nx - 1)):
            buf = _view(ptr, nb); buf[:] = 0
            rec = buf[lenb:].view(np.uint32).reshape(planes, self.ny)
            for j in range(self.ny):
                iv = self._intervals(row * self.ny + j, ln, mat, zp, marks)
                buf[j] = len(iv)
                for p, (z1, m, fl) in enumerate(iv[:planes]):
                    rec[p, j] = z1 | (m << 16) | (fl << 24)

    def _ghost_columns(self, planes, ptr):
        """-> per j: [(z0, z1, mat, marked)] of the layers (gaps dropped), or None without a neighbour."""
        if not ptr:
            return None
        nb = self.ghost_bytes(planes); lenb = (self.ny + 3) & ~3
        buf = _view(ptr, nb)
        rec = buf[lenb:].view(np.uint32).reshape(planes, self.ny)
        cols = []
        for j in range(self.ny):
            prev, lay = 0, []
            for p in range(min(int(buf[j]), planes)):
                r = int(rec[p, j]); z1 = r & 0xFFFF; fl = r >> 24
                if not (fl & 1):
                    lay.append((prev, z1, (r >> 16) & 255, bool(fl & 4)))
                prev = z1
            cols.append(lay)
        return cols

    def _neighbour(self, i, j, own, lo, up):
        """Layers [(z0, z1, mat, marked)] of column (i, j), or None beyond the domain."""
        if j < 0 or j >= self.ny:
            return None
        if 0 <= i < self.nx:
            return own[i * self.ny + j]
        if i == -1 and lo is not None:
            return lo[j]
        if i == self.nx and up is not None:
            return up[j]
        return None

    def _own_columns(self):
        ln, mat, zp, L = self._layers()
        return [[(int(zp[c, k, 0]), int(zp[c, k, 1]), int(mat[c, k]), bool(self._marks[c, k])) for k in range(ln[c])] for c in range(ln.size)]

    def strip_seed(self, matId, planes, ghost_lower, ghost_upper):
        ln, mat, zp, L = self._layers()
        self._marks = np.zeros((ln.size, L), bool)
        own = self._own_columns()
        lo, up = self._ghost_columns(planes, ghost_lower), self._ghost_columns(planes, ghost_upper)
        for i in range(self.nx):
            for j in range(self.ny):
                c = i * self.ny + j
                for k, (z0, z1, m, _) in enumerate(own[c]):
                    if m != matId or z0 == 0:
                        continue
                    exposed = z1 == own[c][-1][1]
                    if not exposed:
                        for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                            nb = self._neighbour(i + di, j + dj, own, lo, up)
                            if nb is None:
                                break
                            if not nb or nb[-1][1] < z1:
                                exposed = True
                                break
                    self._marks[c, k] = exposed

    def strip_sweep(self, matId, planes, ghost_lower, ghost_upper):
        lo, up = self._ghost_columns(planes, ghost_lower), self._ghost_columns(planes, ghost_upper)
        changed_any = 0
        while True:
            own = self._own_columns()
            changed = False
            for i in range(self.nx):
                for j in range(self.ny):
                    c = i * self.ny + j
                    for k, (z0, z1, m, marked) in enumerate(own[c]):
                        if m != matId or marked:
                            continue
                        for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                            nb = self._neighbour(i + di, j + dj, own, lo, up)
                            if nb and any(bm == matId and bk and z0 < b1 and b0 < z1 for (b0, b1, bm, bk) in nb):
                                self._marks[c, k] = True; changed = True
                                break
            if not changed:
                break
            changed_any = 1
        return changed_any

    def strip_finish(self):
        ln, mat, zp, L = self._layers()
        nm = np.zeros_like(mat); nz = np.zeros_like(zp); nl = np.zeros_like(ln)
        for c in range(ln.size):
            w = 0
            for k in range(ln[c]):
                if not self._marks[c, k]:
                    nm[c, w] = mat[c, k]; nz[c, w] = zp[c, k]; w += 1
            nl[c] = w
        self.o.upload(nl.astype(np.uint8), nm.reshape(-1), nz.astype(np.uint16).reshape(-1), L)
        self._marks = None

    def strip_connected(self, matId):
        self.o.strip_connected(matId)

    def identify_cavity(self):
        self.o.identify_cavity()

    def cmp(self, depth, stop=0):
        self.o.cmp(depth, stop)

    def checksum(self):
        return self.o.checksum()

    def maxHeight(self):
        return self.o.max_height()
