"""ctypes binding of the CPU restatement under oracle/ (test infrastructure, never product code)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libcolgrid_oracle.so")

NSTAT = 8


class Shape(C.Structure):
    """cg_shape of include/colgrid.h."""
    _fields_ = [("type", C.c_int32), ("positive", C.c_int32), ("rotated", C.c_int32),
                ("pt_off", C.c_int32), ("pt_cnt", C.c_int32), ("_pad", C.c_int32),
                ("left", C.c_double), ("top", C.c_double), ("w", C.c_double), ("h", C.c_double),
                ("cosr", C.c_double), ("sinr", C.c_double), ("scale_x", C.c_double),
                ("scale_y", C.c_double), ("a", C.c_double), ("b", C.c_double)]


def shapes_from_reduced(reduced):
    """reduced = {'shapes': [[15 numbers]...], 'pts': [...]} as written by maskshapes.js."""
    n = len(reduced["shapes"])
    arr = (Shape * max(1, n))()
    for s, r in enumerate(reduced["shapes"]):
        (arr[s].type, arr[s].positive, arr[s].rotated, arr[s].pt_off, arr[s].pt_cnt) = [int(v) for v in r[:5]]
        (arr[s].left, arr[s].top, arr[s].w, arr[s].h, arr[s].cosr, arr[s].sinr,
         arr[s].scale_x, arr[s].scale_y, arr[s].a, arr[s].b) = [float(v) for v in r[5:15]]
    pts = np.asarray(reduced["pts"], dtype=np.float64)
    return arr, n, pts


def _lib():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("colgrid_oracle.c", "colgrid_oracle_ald_wet.inc")]
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = C.CDLL(_SO)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.og_create.restype = vp; L.og_create.argtypes = [d, d, d, d, i, d]
    L.og_create_slab.restype = vp; L.og_create_slab.argtypes = [i, i, i, i, d, d, i, d]
    L.og_destroy.argtypes = [vp]
    L.og_reset.argtypes = [vp, i]
    for f in ("og_nx", "og_ny", "og_lmax"):
        getattr(L, f).argtypes = [vp]
    L.og_mat.restype = C.POINTER(C.c_uint8); L.og_mat.argtypes = [vp]
    L.og_zpair.restype = C.POINTER(C.c_uint16); L.og_zpair.argtypes = [vp]
    L.og_len.restype = C.POINTER(C.c_uint8); L.og_len.argtypes = [vp]
    L.og_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.og_upload.argtypes = [vp, vp, vp, vp, i]
    L.og_deposit_general.argtypes = [vp, vp, i, d, d]
    L.og_etch_general.argtypes = [vp, vp, i, d, d]
    L.og_deposit_general_halo.argtypes = [vp, vp, i, d, d, vp, vp]
    L.og_etch_general_halo.argtypes = [vp, vp, i, d, d, vp, vp, vp, vp]
    L.og_etch_flags.argtypes = [vp, vp, i, vp]
    L.og_top_heights.argtypes = [vp, vp]
    L.og_max_height.restype = d; L.og_max_height.argtypes = [vp]
    L.og_strip_connected.argtypes = [vp, i]
    L.og_identify_cavity.argtypes = [vp]
    L.og_cmp.argtypes = [vp, d, i]
    L.og_add_layer.argtypes = [vp, i, i, i, d, d]
    L.og_remove_top_layer.argtypes = [vp, i, i]
    L.og_mask_raster.argtypes = [vp, vp, i, vp, vp]
    L.og_checksum.restype = C.c_uint64; L.og_checksum.argtypes = [vp]
    L.og_fnv.restype = C.c_uint32; L.og_fnv.argtypes = [vp]
    L.og_kernel_info.argtypes = [vp, d, d, C.POINTER(i), C.POINTER(i)]
    L.og_kernel_taps.argtypes = [vp, d, d, vp, vp, vp]
    L.og_index_from_pos.argtypes = [vp, i, d]
    L.og_etch_wet.argtypes = [vp, i, d, d, i]
    L.og_etch_wet_step.argtypes = [vp, i, i, i]
    L.og_etch_wet_step_part.argtypes = [vp, i, i, i, i]
    L.og_deposit_ald.argtypes = [vp, i, d, d, i]
    L.og_slider_reset.argtypes = [vp]
    L.og_slider_ald_entries.argtypes = [vp]
    L.og_slider_wet_entries.argtypes = [vp]
    L.og_slider_seed_count.restype = C.c_size_t; L.og_slider_seed_count.argtypes = [vp]
    L.og_ald_taps.argtypes = [vp, d, vp, vp, vp]
    L.og_ald_seeds_part.restype = C.c_size_t; L.og_ald_seeds_part.argtypes = [vp, d, d, vp, vp, vp, vp, C.c_size_t]
    L.og_ald_coat_part.argtypes = [vp, i, d, d, vp, vp, vp, vp, C.c_size_t]
    L.og_ald_merge_part.argtypes = [vp, d, vp, vp, vp, i]
    return L


_L = None


def lib():
    global _L
    if _L is None:
        _L = _lib()
    return _L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleGrid:
    """Mirror of the reference ColumnGrid surface on top of oracle/colgrid_oracle.c."""

    def __init__(self, LX=None, LY=None, dx=2.0, dy=2.0, lmax=32, zscale=0.1, slab=None):
        self.L = lib()
        if slab is not None:
            nxg, ny, i0, nxl = slab
            self.h = self.L.og_create_slab(nxg, ny, i0, nxl, dx, dy, lmax, zscale)
        else:
            self.h = self.L.og_create(LX, LY, dx, dy, lmax, zscale)
        assert self.h
        self.dx, self.dy = dx, dy

    def __del__(self):
        if getattr(self, "h", None):
            self.L.og_destroy(self.h); self.h = None

    NX = property(lambda s: s.L.og_nx(s.h))
    NY = property(lambda s: s.L.og_ny(s.h))
    Lmax = property(lambda s: s.L.og_lmax(s.h))

    def _mask(self, mask):
        if mask is None:
            return None
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        assert m.size == self.NX * self.NY
        return m

    def deposit_general(self, mask, mat, t, S):
        m = self._mask(mask); self.L.og_deposit_general(self.h, _ptr(m), mat, t, S)

    def etch_general(self, mask, mat, t, S):
        m = self._mask(mask); self.L.og_etch_general(self.h, _ptr(m), mat, t, S)

    def deposit_general_halo(self, mask, mat, t, S, lowerH, upperH):
        m = self._mask(mask)
        self.L.og_deposit_general_halo(self.h, _ptr(m), mat, t, S, _ptr(lowerH), _ptr(upperH))

    def etch_general_halo(self, mask, mat, t, S, lowerH, upperH, lowerE, upperE):
        m = self._mask(mask)
        self.L.og_etch_general_halo(self.h, _ptr(m), mat, t, S, _ptr(lowerH), _ptr(upperH),
                                    _ptr(lowerE), _ptr(upperE))

    def etch_flags(self, mask, mat):
        m = self._mask(mask); E = np.zeros(self.NX * self.NY, np.uint8)
        self.L.og_etch_flags(self.h, _ptr(m), mat, _ptr(E)); return E.reshape(self.NX, self.NY)

    def top_heights(self):
        H = np.zeros(self.NX * self.NY, np.float32)
        self.L.og_top_heights(self.h, _ptr(H)); return H.reshape(self.NX, self.NY)

    def cmp(self, depth, stop):
        self.L.og_cmp(self.h, depth, stop)

    def strip_connected(self, mat):
        self.L.og_strip_connected(self.h, mat)

    def identify_cavity(self):
        self.L.og_identify_cavity(self.h)

    def reset(self, lmax=0):
        self.L.og_reset(self.h, lmax)

    def addLayer(self, i, j, mat, z0, z1):
        self.L.og_add_layer(self.h, i, j, mat, z0, z1)

    def removeTopLayer(self, i, j):
        self.L.og_remove_top_layer(self.h, i, j)

    def etch_wet(self, mat, t, dt=None, isCache=False):
        rc = self.L.og_etch_wet(self.h, mat, t, float("nan") if dt is None else dt, int(bool(isCache)))
        if rc:
            raise RuntimeError("the reference throws here (etch_wet rc=%d)" % rc)

    def etch_wet_step(self, mat, dtq, diagonal):
        rc = self.L.og_etch_wet_step(self.h, mat, dtq, int(bool(diagonal)))
        if rc:
            raise RuntimeError("the reference throws here (etch_wet step rc=%d)" % rc)

    def etch_wet_step_part(self, mat, dtq, diagonal, hq):
        rc = self.L.og_etch_wet_step_part(self.h, mat, dtq, int(bool(diagonal)), int(hq))
        if rc:
            raise RuntimeError("the reference throws here (etch_wet step rc=%d)" % rc)

    def deposit_ALD(self, mat, t, dt=None, isCache=False):
        rc = self.L.og_deposit_ald(self.h, mat, t, float("nan") if dt is None else dt, int(bool(isCache)))
        if rc:
            raise RuntimeError("the reference loses its state here (deposit_ALD rc=%d)" % rc)

    def slider_reset(self):
        self.L.og_slider_reset(self.h)

    def slider_sizes(self):
        return (self.L.og_slider_ald_entries(self.h), self.L.og_slider_wet_entries(self.h),
                self.L.og_slider_seed_count(self.h))

    def ald_taps(self, t):
        n = self.L.og_ald_taps(self.h, t, None, None, None)
        du = np.zeros(n, np.int32); dv = np.zeros(n, np.int32); add = np.zeros(n, np.float64)
        self.L.og_ald_taps(self.h, t, _ptr(du), _ptr(dv), _ptr(add))
        return du, dv, add

    def max_height(self):
        return self.L.og_max_height(self.h)

    # deposit_ALD in the pieces one slab of a larger domain runs; h_nm = the whole domain's height
    def ald_seeds_part(self, T, h_nm):
        """-> (i, j, zAq, zBq) arrays of the seeds collectALDSeeds finds on this grid."""
        n = self.L.og_ald_seeds_part(self.h, T, h_nm, None, None, None, None, 0)
        si = np.zeros(max(n, 1), np.int32); sj = np.zeros(max(n, 1), np.int32)
        za = np.zeros(max(n, 1), np.uint32); zb = np.zeros(max(n, 1), np.uint32)
        self.L.og_ald_seeds_part(self.h, T, h_nm, _ptr(si), _ptr(sj), _ptr(za), _ptr(zb), n)
        return si[:n], sj[:n], za[:n], zb[:n]

    def ald_coat_part(self, mat, t, h_nm, si, sj, za, zb):
        si = np.ascontiguousarray(si, np.int32); sj = np.ascontiguousarray(sj, np.int32)
        za = np.ascontiguousarray(za, np.uint32); zb = np.ascontiguousarray(zb, np.uint32)
        rc = self.L.og_ald_coat_part(self.h, int(mat), float(t), float(h_nm), _ptr(si), _ptr(sj), _ptr(za), _ptr(zb), si.size)
        if rc:
            raise RuntimeError("the reference throws here (deposit_ALD coat rc=%d)" % rc)

    def ald_merge_part(self, hbase_nm, ln, mat, zp, lmax):
        ln = np.ascontiguousarray(ln, np.uint8); mat = np.ascontiguousarray(mat, np.uint8); zp = np.ascontiguousarray(zp, np.uint16)
        rc = self.L.og_ald_merge_part(self.h, float(hbase_nm), _ptr(ln), _ptr(mat), _ptr(zp), int(lmax))
        if rc:
            raise RuntimeError("the reference throws here (deposit_ALD merge rc=%d)" % rc)

    def checksum(self):
        return self.L.og_checksum(self.h)

    def fnv(self):
        return "%08x" % self.L.og_fnv(self.h)

    def stats(self):
        out = (C.c_uint64 * NSTAT)(); self.L.og_stats(self.h, out); return list(out)

    def arrays(self):
        """(len, mat, zpair) copies in the reference layout."""
        N, Lm = self.NX * self.NY, self.Lmax
        ln = np.ctypeslib.as_array(self.L.og_len(self.h), (N,)).copy()
        mat = np.ctypeslib.as_array(self.L.og_mat(self.h), (N * Lm,)).copy()
        zp = np.ctypeslib.as_array(self.L.og_zpair(self.h), (2 * N * Lm,)).copy()
        return ln, mat, zp

    def upload(self, ln, mat, zp, lmax):
        ln = np.ascontiguousarray(ln, np.uint8); mat = np.ascontiguousarray(mat, np.uint8)
        zp = np.ascontiguousarray(zp, np.uint16)
        self.L.og_upload(self.h, _ptr(mat), _ptr(zp), _ptr(ln), lmax)

    def mask_raster(self, reduced):
        arr, n, pts = shapes_from_reduced(reduced)
        out = np.zeros(self.NX * self.NY, np.uint8)
        self.L.og_mask_raster(self.h, C.cast(arr, C.c_void_p), n, _ptr(pts) if pts.size else None, _ptr(out))
        return out

    def kernel_taps(self, t, S):
        Rx, n = C.c_int(), C.c_int()
        self.L.og_kernel_info(self.h, t, S, C.byref(Rx), C.byref(n))
        du = np.zeros(n.value, np.int32); dv = np.zeros(n.value, np.int32); add = np.zeros(n.value, np.float32)
        self.L.og_kernel_taps(self.h, t, S, _ptr(du), _ptr(dv), _ptr(add))
        return Rx.value, du, dv, add

    def index_from_pos(self, axis, pos):
        return self.L.og_index_from_pos(self.h, axis, pos)
