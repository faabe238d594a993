"""CPU stand-in for one rank's grid in the slab driver tests (test infrastructure only).

It offers the same stencil_begin / stencil_finish / set_halo_buffers surface as
fem_amd.ColumnGrid, implemented with the CPU oracle, and speaks the same halo wire format
(R*ny u16 top heights followed by R*ny u8 etchable flags per direction)."""
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

    def checksum(self):
        return self.o.checksum()

    def maxHeight(self):
        return self.o.max_height()
