"""ctypes view of libcolgrid's C ABI (include/colgrid.h) with the reference's method names.

`ColumnGrid` mirrors the call surface of the reference's ColumnGrid class
(template/script_columngrid.js:12-43, 435, 598, 747, 867; script_columngrid2.js:9, 85):
same method names, argument order and "bad input is a silent no-op" behaviour.  The one
difference is the mask argument: the reference takes a JavaScript closure maskFn(x, y); here a
mask is a device raster built once per mask (`Mask`), `None` meaning "no mask".

There is no CPU fallback: if the HIP library cannot be loaded every constructor raises.
"""
import ctypes as C
import array
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COLGRID_LIB") or os.path.join(_HERE, "lib", "libcolgrid.so")      # the override is for A/B timing of two builds

CG_MASK_NONE = -1
CG_NSTAT = 8
CG_NKIND = 12
KERNEL_KINDS = ["k_vertical_g4<deposit>", "k_vertical_g4<etch>", "k_vertical_fused", "k_halo_pack / k_ghost_pack",
                "k_stencil<deposit>", "k_stencil<etch>", "k_mask_shapes", "clear(memset)",
                "k_max_top", "other", "k_stencil<deposit+vertical>", "-"]      # k_stencil = k_stencil_stream (k_stencil_sym / k_stencil_apply where it does not apply)
STENCIL_DEPOSIT, STENCIL_ETCH = 0, 1
SHAPE_RECT, SHAPE_CIRCLE, SHAPE_ELLIPSE, SHAPE_POLYGON, SHAPE_NEVER = range(5)


class ColGridError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcolgrid error {code}: {msg}")
        self.code = code


class CgInfo(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nx_global", C.c_int32), ("i0", C.c_int32),
                ("lmax", C.c_int32), ("device", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("zscale", C.c_double),
                ("offset_x", C.c_double), ("offset_y", C.c_double), ("lx_eff", C.c_double), ("ly_eff", C.c_double)]


class CgShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("positive", C.c_int32), ("rotated", C.c_int32),
                ("pt_off", C.c_int32), ("pt_cnt", C.c_int32), ("_pad", C.c_int32),
                ("left", C.c_double), ("top", C.c_double), ("w", C.c_double), ("h", C.c_double),
                ("cosr", C.c_double), ("sinr", C.c_double), ("scale_x", C.c_double), ("scale_y", C.c_double),
                ("a", C.c_double), ("b", C.c_double)]


# every symbol include/colgrid.h declares, with its ctypes signature
_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_pp = C.POINTER(C.c_void_p)
ABI = {
    "cg_create": (_i, [_d, _d, _d, _d, _i, _d, _pp]),
    "cg_create_slab": (_i, [_i, _i, _i, _i, _d, _d, _i, _d, _pp]),
    "cg_destroy": (_i, [_vp]),
    "cg_reset": (_i, [_vp, _i]),
    "cg_clear": (_i, [_vp]),
    "cg_raise_lmax": (_i, [_vp, _i]),
    "cg_get_info": (_i, [_vp, C.POINTER(CgInfo)]),
    "cg_last_error": (C.c_char_p, [_vp]),
    "cg_set_stream": (_i, [_vp, _vp]),
    "cg_get_stream": (_i, [_vp, _pp]),
    "cg_sync": (_i, [_vp]),
    "cg_mask_from_raster": (_i, [_vp, _vp, _sz, C.POINTER(_i)]),
    "cg_mask_from_device_raster": (_i, [_vp, _vp, _sz, C.POINTER(_i)]),
    "cg_mask_from_shapes": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(_i)]),
    "cg_mask_any_open": (_i, [_vp, _i, C.POINTER(_i)]),
    "cg_kernel_cache_touch": (_i, [_vp, _d, _d]),
    "cg_mask_download": (_i, [_vp, _i, _vp, _sz]),
    "cg_mask_destroy": (_i, [_vp, _i]),
    "cg_deposit_general": (_i, [_vp, _i, _i, _d, _d]),
    "cg_etch_general": (_i, [_vp, _i, _i, _d, _d]),
    "cg_cmp": (_i, [_vp, _d, _i]),
    "cg_strip_connected": (_i, [_vp, _i]),
    "cg_identify_cavity": (_i, [_vp]),
    "cg_cmp_scan": (_i, [_vp, _i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_i)]),
    "cg_cmp_apply": (_i, [_vp, _d, _i]),
    "cg_cavity_state": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_uint32)]),
    "cg_cavity_fill": (_i, [_vp, C.c_uint32]),
    "cg_ghost_bytes": (_i, [_vp, _i, C.POINTER(_sz)]),
    "cg_ghost_pack": (_i, [_vp, _i, _vp, _vp]),
    "cg_strip_seed": (_i, [_vp, _i, _i, _vp, _vp]),
    "cg_strip_sweep": (_i, [_vp, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "cg_strip_finish": (_i, [_vp]),
    "cg_etch_wet": (_i, [_vp, _i, _d, _d]),
    "cg_etch_wet_step": (_i, [_vp, _i, _i, _i]),
    "cg_etch_wet_step_slab": (_i, [_vp, _i, _i, _i, C.c_uint32, _i, _vp, _vp]),
    "cg_deposit_ald": (_i, [_vp, _i, _d, _d]),
    "cg_ald_collect_seeds": (_i, [_vp, _d, C.POINTER(C.c_uint64)]),
    "cg_ald_seed_info": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_uint64)]),
    "cg_ald_drop_seeds": (_i, [_vp]),
    "cg_ald_increment": (_i, [_vp, _i, _d]),
    "cg_ald_coat": (_i, [_vp, _i, _d]),
    "cg_ald_collect_seeds_slab": (_i, [_vp, _d, C.c_uint32, _i, _vp, _vp, C.POINTER(_i), C.POINTER(C.c_uint64)]),
    "cg_ald_seed_halo_bytes": (_i, [_vp, _i, C.POINTER(_sz)]),
    "cg_ald_seed_halo_pack": (_i, [_vp, _i, _vp, _vp]),
    "cg_ald_seed_halo_unpack": (_i, [_vp, _i, _vp, _vp]),
    "cg_ald_coat_slab": (_i, [_vp, _i, _d, C.c_uint32]),
    "cg_ald_merge_slab": (_i, [_vp, C.c_uint32]),
    "cg_release_scratch": (_i, [_vp]),
    "cg_set_plane_budget": (_i, [_vp, _i]),
    "cg_add_layer": (_i, [_vp, _i, _i, _i, _d, _d]),
    "cg_remove_top_layer": (_i, [_vp, _i, _i]),
    "cg_max_height": (_i, [_vp, C.POINTER(_d), C.POINTER(C.c_uint32)]),
    "cg_stencil_begin": (_i, [_vp, _i, _i, _i, _d, _d, _i, _i, C.POINTER(_i)]),
    "cg_halo_buffers": (_i, [_vp, _pp, _pp, _pp, _pp, C.POINTER(_sz)]),
    "cg_stencil_finish": (_i, [_vp]),
    "cg_stencil_interior": (_i, [_vp]),
    "cg_set_halo_buffers": (_i, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "cg_profile": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_uint64)]),
    "cg_download": (_i, [_vp, _vp, _vp, _vp]),
    "cg_upload": (_i, [_vp, _vp, _vp, _vp, _i]),
    "cg_snapshot_save": (_i, [_vp, _i]),
    "cg_snapshot_load": (_i, [_vp, _i]),
    "cg_snapshot_drop": (_i, [_vp, _i]),
    "cg_slices": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "cg_slices_device": (_i, [_vp, _i, _vp, _i, _pp, _pp, _pp]),
    "cg_slices_noalloc": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "cg_index_from_pos": (_i, [_vp, _i, _d, C.POINTER(C.c_int32)]),
    "cg_checksum": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "cg_step_stats": (_i, [_vp, _i, C.POINTER(C.c_uint64)]),
    "cg_timer_start": (_i, [_vp]),
    "cg_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "cg_version": (C.c_char_p, []),
    # a grid cut into slabs that the library drives itself (cg_group)
    "cg_group_create": (_i, [_i, _i, _d, _d, _i, _d, C.POINTER(_i), _i, _pp]),
    "cg_group_destroy": (_i, [_vp]),
    "cg_group_last_error": (C.c_char_p, [_vp]),
    "cg_group_slab_count": (_i, [_vp, C.POINTER(_i)]),
    "cg_group_slab": (_i, [_vp, _i, _pp]),
    "cg_group_transport": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cg_group_get_info": (_i, [_vp, C.POINTER(CgInfo)]),
    "cg_group_sync": (_i, [_vp]),
    "cg_group_reset": (_i, [_vp, _i]),
    "cg_group_clear": (_i, [_vp]),
    "cg_group_set_plane_budget": (_i, [_vp, _i]),
    "cg_group_mask_from_shapes": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(_i)]),
    "cg_group_mask_from_raster": (_i, [_vp, _vp, _sz, C.POINTER(_i)]),
    "cg_group_mask_download": (_i, [_vp, _i, _vp, _sz]),
    "cg_group_mask_destroy": (_i, [_vp, _i]),
    "cg_group_deposit_general": (_i, [_vp, _i, _i, _d, _d]),
    "cg_group_etch_general": (_i, [_vp, _i, _i, _d, _d]),
    "cg_group_cmp": (_i, [_vp, _d, _i]),
    "cg_group_strip_connected": (_i, [_vp, _i]),
    "cg_group_identify_cavity": (_i, [_vp]),
    "cg_group_etch_wet": (_i, [_vp, _i, _d, _d]),
    "cg_group_etch_wet_step": (_i, [_vp, _i, _i, _i]),
    "cg_group_deposit_ald": (_i, [_vp, _i, _d, _d]),
    "cg_group_ald_collect_seeds": (_i, [_vp, _d, C.POINTER(C.c_uint64)]),
    "cg_group_ald_drop_seeds": (_i, [_vp]),
    "cg_group_ald_increment": (_i, [_vp, _i, _d]),
    "cg_group_ald_coat": (_i, [_vp, _i, _d]),
    "cg_group_release_scratch": (_i, [_vp]),
    "cg_group_add_layer": (_i, [_vp, _i, _i, _i, _d, _d]),
    "cg_group_remove_top_layer": (_i, [_vp, _i, _i]),
    "cg_group_max_height": (_i, [_vp, C.POINTER(_d), C.POINTER(C.c_uint32)]),
    "cg_group_checksum": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "cg_group_download": (_i, [_vp, _vp, _vp, _vp]),
    "cg_group_upload": (_i, [_vp, _vp, _vp, _vp, _i]),
    "cg_group_snapshot_save": (_i, [_vp, _i]),
    "cg_group_snapshot_load": (_i, [_vp, _i]),
    "cg_group_snapshot_drop": (_i, [_vp, _i]),
    "cg_group_slices": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "cg_group_index_from_pos": (_i, [_vp, _i, _d, C.POINTER(C.c_int32)]),
}

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.

    PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and opens them by path.  If
    libcolgrid pulled in the system runtime first, a later `import torch` would bring a second
    runtime into the process and whichever initialises second sees no GPU.  So when torch is
    installed its runtime is loaded first (without importing torch); libcolgrid's NEEDED entry
    then binds to it by soname.  Processes without torch (e.g. the Node addon) use /opt/rocm.
    """
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def load_library(path=None):
    """Load libcolgrid.so and bind every ABI symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ColGridError(-1, f"{p} not found: build it with `make -C fem_amd/csrc` (hipcc, gfx950)")
    _preload_torch_hip_runtime()
    lib = C.CDLL(p)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


SHAPE_DTYPE = np.dtype([("type", "<i4"), ("positive", "<i4"), ("rotated", "<i4"), ("pt_off", "<i4"), ("pt_cnt", "<i4"), ("_pad", "<i4"),
                        ("left", "<f8"), ("top", "<f8"), ("w", "<f8"), ("h", "<f8"), ("cosr", "<f8"), ("sinr", "<f8"),
                        ("scale_x", "<f8"), ("scale_y", "<f8"), ("a", "<f8"), ("b", "<f8")])      # = CgShape, field for field
_SHAPE_FIELDS = [n for n in SHAPE_DTYPE.names if n != "_pad"]
assert SHAPE_DTYPE.itemsize == C.sizeof(CgShape)


def reduce_mask(mask_data, packed=False):
    """Python twin of fem_amd/js/maskshapes.js: fabric JSON -> {'open', 'shapes', 'pts'}.

    Follows template/script_runtime.js:54-129 of the reference.  cos/sin come from this
    process's libm; a JavaScript host passes its own values instead.
    packed=True: 'shapes' is a numpy record array laid out as cg_shape (what mask_from_shapes hands to
    the library as it is: masks of 10^5 shapes and more then cost no second pass over Python objects).
    """
    if not mask_data or not mask_data.get("objects"):
        if not mask_data or "objects" not in mask_data or mask_data["objects"] is None:
            return {"open": True, "shapes": np.zeros(0, SHAPE_DTYPE) if packed else [], "pts": []}
    shapes, pts = [], []
    flat = array.array("d")
    add = flat.extend if packed else None
    for obj in mask_data["objects"]:
        if not obj.get("visible"):
            continue
        left, top = obj.get("left") or 0, obj.get("top") or 0
        w, h = obj.get("width") or 0, obj.get("height") or 0
        angle = obj.get("angle") or 0
        sx, sy = obj.get("scaleX") or 1, obj.get("scaleY") or 1
        polarity = (obj.get("data") or {}).get("polarity") or "positive"
        positive = 1 if polarity == "positive" else 0
        rotated = 1 if angle != 0 else 0
        cosr, sinr = 1.0, 0.0
        if rotated:
            rad = (-angle * math.pi) / 180
            cosr, sinr = math.cos(rad), math.sin(rad)
        typ, a, b, off, cnt = SHAPE_NEVER, 0.0, 0.0, 0, 0
        t = obj.get("type")
        if t == "rect":
            typ = SHAPE_RECT
        elif t == "circle":
            typ, a = SHAPE_CIRCLE, (obj.get("radius") or w / 2)
        elif t == "ellipse":
            typ, a, b = SHAPE_ELLIPSE, (obj.get("rx") or w / 2), (obj.get("ry") or h / 2)
        elif t == "polygon":
            if obj.get("points") is None:
                raise TypeError("polygon without points")
            typ, off, cnt = SHAPE_POLYGON, len(pts) // 2, len(obj["points"])
            for p in obj["points"]:
                pts += [p["x"], p["y"]]
        elif t == "path" and obj.get("path"):
            typ, off = SHAPE_POLYGON, len(pts) // 2
            for p in obj["path"]:
                if p[0] in ("L", "M"):
                    pts += [p[1], p[2]]; cnt += 1
        if packed:
            try:
                add((typ, positive, rotated, off, cnt, left, top, w, h, cosr, sinr, sx, sy, a, b))
            except TypeError:                 # numbers given as text and the like
                del flat[15 * (len(flat) // 15):]
                add([float(v) for v in (typ, positive, rotated, off, cnt, left, top, w, h, cosr, sinr, sx, sy, a, b)])
        else:
            shapes.append([typ, positive, rotated, off, cnt, left, top, w, h, cosr, sinr, sx, sy, a, b])
    if packed:
        shapes = _records(np.frombuffer(flat, dtype=np.float64).reshape(-1, 15))
    return {"open": False, "shapes": shapes, "pts": pts}


def _records(table):
    """(n, 15) numbers in the order of a reduced shape -> n records of SHAPE_DTYPE."""
    rec = np.zeros(table.shape[0], SHAPE_DTYPE)
    for k, name in enumerate(_SHAPE_FIELDS):
        rec[name] = table[:, k]
    return rec


def _pack_shapes(reduced):
    """-> (record array kept alive by the caller, number of shapes, points as float64)."""
    sh = reduced["shapes"]
    if not (isinstance(sh, np.ndarray) and sh.dtype == SHAPE_DTYPE):
        sh = _records(np.asarray(sh, dtype=np.float64).reshape(-1, 15))
    sh = np.ascontiguousarray(sh)
    pts = np.ascontiguousarray(reduced["pts"], dtype=np.float64)
    return sh, int(sh.shape[0]), pts


class Mask:
    """A device raster of one mask for one grid."""

    def __init__(self, grid, mask_id):
        self.grid, self.id = grid, mask_id

    def download(self):
        out = np.empty(self.grid.NX * self.grid.NY, np.uint8)
        self.grid._chk(self.grid.lib.cg_mask_download(self.grid.h, self.id, out.ctypes.data, out.size))
        return out

    def destroy(self):
        if self.id is not None and self.grid.h:
            self.grid.lib.cg_mask_destroy(self.grid.h, self.id)
        self.id = None


class ColumnGrid:
    """The reference's ColumnGrid call surface over libcolgrid.

    ColumnGrid(LXnm, LYnm, dx, dy, LmaxInit=32, zScale=0.1) as script_columngrid.js:21.
    slab=(nx_global, ny, i0, nx_local) makes a slab of a larger domain instead (multi-GPU).
    """

    def __init__(self, LXnm=None, LYnm=None, dx=2.0, dy=2.0, LmaxInit=32, zScale=0.1, *, slab=None, lib=None):
        self.lib = lib or load_library()
        self.h = None
        out = C.c_void_p()
        if slab is not None:
            nxg, ny, i0, nxl = slab
            rc = self.lib.cg_create_slab(nxg, ny, i0, nxl, dx, dy, LmaxInit, zScale, C.byref(out))
        else:
            rc = self.lib.cg_create(LXnm, LYnm, dx, dy, LmaxInit, zScale, C.byref(out))
        if rc:
            raise ColGridError(rc, (self.lib.cg_last_error(None) or b"").decode())
        self.h = out
        self.nMaxCache = 10
        self._info = None
        # sliderCache of the reference (script_columngrid.js:34): anchors of etch_wet / deposit_ALD
        self.sliderCache = {}
        self.keep_step_snapshots = True      # the reference stores one grid copy per increment
        self._next_slot = 1 << 24

    # ---- plumbing
    def _chk(self, rc):
        if rc:
            raise ColGridError(rc, (self.lib.cg_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.lib.cg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        inf = CgInfo()
        self._chk(self.lib.cg_get_info(self.h, C.byref(inf)))
        return inf

    NX = property(lambda s: s.info().nx)
    NY = property(lambda s: s.info().ny)
    Lmax = property(lambda s: s.info().lmax)
    dx = property(lambda s: s.info().dx)
    dy = property(lambda s: s.info().dy)
    offsetX = property(lambda s: s.info().offset_x)
    offsetY = property(lambda s: s.info().offset_y)
    LXeff = property(lambda s: s.info().lx_eff)
    LYeff = property(lambda s: s.info().ly_eff)

    def sync(self):
        self._chk(self.lib.cg_sync(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.cg_set_stream(self.h, stream_ptr))

    def get_stream(self):
        """The hipStream_t (as an int) the handle's work is enqueued on."""
        p = C.c_void_p()
        self._chk(self.lib.cg_get_stream(self.h, C.byref(p)))
        return p.value or 0

    # ---- masks
    def mask_from_raster(self, raster):
        r = np.ascontiguousarray(raster, dtype=np.uint8).reshape(-1)
        mid = C.c_int()
        self._chk(self.lib.cg_mask_from_raster(self.h, r.ctypes.data, r.size, C.byref(mid)))
        return Mask(self, mid.value)

    def mask_from_device_raster(self, dev_ptr, n):
        mid = C.c_int()
        self._chk(self.lib.cg_mask_from_device_raster(self.h, dev_ptr, n, C.byref(mid)))
        return Mask(self, mid.value)

    def mask_from_shapes(self, reduced):
        """reduced: output of reduce_mask() / maskshapes.js reduceMask()."""
        arr, n, pts = _pack_shapes(reduced)
        mid = C.c_int()
        self._chk(self.lib.cg_mask_from_shapes(self.h, arr.ctypes.data if n else None, n,
                                               pts.ctypes.data if pts.size else None, pts.size // 2, C.byref(mid)))
        return Mask(self, mid.value)

    def mask_any_open(self, mask):
        """Does the mask open any column of this handle (None: yes)?"""
        a = C.c_int()
        self._chk(self.lib.cg_mask_any_open(self.h, self._mid(mask), C.byref(a)))
        return bool(a.value)

    def kernel_cache_touch(self, thickness, conformality):
        """The conformal-kernel cache gets the entry a lateral step of these parameters would create
        (script_columngrid.js:470), without anything else happening."""
        self._chk(self.lib.cg_kernel_cache_touch(self.h, float(thickness), float(conformality)))

    def mask_from_data(self, mask_data):
        red = reduce_mask(mask_data)
        return None if red["open"] else self.mask_from_shapes(red)

    @staticmethod
    def _mid(mask):
        if mask is None:
            return CG_MASK_NONE
        if isinstance(mask, Mask):
            return mask.id
        raise TypeError("mask must be None or a Mask made by this grid (mask_from_raster / mask_from_shapes)")

    # ---- the reference's step methods
    def deposit_general(self, mask, matId, thickness, conformality=1.0):
        self._chk(self.lib.cg_deposit_general(self.h, self._mid(mask), int(matId), float(thickness), float(conformality)))

    def etch_general(self, mask, matId, thickness, conformality=1.0):
        self._chk(self.lib.cg_etch_general(self.h, self._mid(mask), int(matId), float(thickness), float(conformality)))

    def cmp(self, depth, stopMatId=0):
        self._chk(self.lib.cg_cmp(self.h, float(depth), int(stopMatId)))

    def strip_connected(self, matId):
        self._chk(self.lib.cg_strip_connected(self.h, int(matId)))

    def identify_cavity(self):
        self._chk(self.lib.cg_identify_cavity(self.h))

    # ---- cmp / strip_connected / identify_cavity in the pieces a slab driver needs (see include/colgrid.h)
    def cmp_scan(self, stopMatId):
        """-> (top_q, stopper_q, stopper_found) of this handle's columns."""
        a, b, f = C.c_uint32(), C.c_uint32(), C.c_int()
        self._chk(self.lib.cg_cmp_scan(self.h, int(stopMatId), C.byref(a), C.byref(b), C.byref(f)))
        return a.value, b.value, f.value

    def cmp_apply(self, target_z, stopMatId):
        self._chk(self.lib.cg_cmp_apply(self.h, float(target_z), int(stopMatId)))

    def cavity_state(self, flags_only=False):
        """-> (has_gaps, has_cavity, max_len, top_q); flags_only skips the reduction behind the last two."""
        a, b, m, q = C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
        if flags_only:
            self._chk(self.lib.cg_cavity_state(self.h, C.byref(a), C.byref(b), None, None))
        else:
            self._chk(self.lib.cg_cavity_state(self.h, C.byref(a), C.byref(b), C.byref(m), C.byref(q)))
        return a.value, b.value, m.value, q.value

    def cavity_fill(self, hmax_q):
        self._chk(self.lib.cg_cavity_fill(self.h, int(hmax_q)))

    def ghost_bytes(self, planes):
        n = C.c_size_t()
        self._chk(self.lib.cg_ghost_bytes(self.h, int(planes), C.byref(n)))
        return n.value

    def ghost_pack(self, planes, send_lower, send_upper):
        self._chk(self.lib.cg_ghost_pack(self.h, int(planes), send_lower, send_upper))

    def strip_seed(self, matId, planes, ghost_lower, ghost_upper):
        self._chk(self.lib.cg_strip_seed(self.h, int(matId), int(planes), ghost_lower or None, ghost_upper or None))

    def strip_sweep(self, matId, planes, ghost_lower, ghost_upper):
        ch = C.c_int()
        self._chk(self.lib.cg_strip_sweep(self.h, int(matId), int(planes), ghost_lower or None, ghost_upper or None, C.byref(ch)))
        return ch.value

    def strip_finish(self):
        self._chk(self.lib.cg_strip_finish(self.h))

    @property
    def zScale(self):
        return self.info().zscale

    # ---- etch_wet / deposit_ALD with the reference's anchors (sliderCache)
    @staticmethod
    def _js_round(x):
        r = math.floor(x)
        return r + 1 if x - r >= 0.5 else r

    def _drop_slots(self, slots):
        for slot in slots:
            self.snapshot_drop(slot)

    def slider_reset(self):
        """this.sliderCache = {} (script_columngrid.js:244, script_runtime.js:275)."""
        wc = self.sliderCache.get("wetetchCache")
        ac = self.sliderCache.get("aldCache")
        if wc:
            self._drop_slots(wc["grid"].values())
        if ac:
            self._drop_slots(ac["gridWithCavity"].values())
            self._chk(self.lib.cg_ald_drop_seeds(self.h))
        self.sliderCache = {}

    def _anchor(self, table, T):
        best = -1
        for k in table:
            if k <= T + 1e-9 and k > best:
                best = k
        return best if best >= 0 else None

    def _save_anchor(self, table, key):
        if not self.keep_step_snapshots:
            return
        if key not in table:
            table[key] = self._next_slot
            self._next_slot += 1
        self.snapshot_save(table[key])

    def etch_wet(self, matId, totalThk, dt=None, isCache=False):
        """etch_wet(matId, totalThk, {dt, isCache}) (script_columngrid2.js:893 - script_columngrid3.js:308)."""
        T = float(totalThk)
        T = 0.0 if T != T else max(0.0, T)
        if T <= 0:
            return
        inf = self.info()
        zs = inf.zscale
        base_dt = max(min(inf.dx, inf.dy) if dt is None else float(dt), 1e-6)
        wc = self.sliderCache.setdefault("wetetchCache", {"grid": {}, "R": 0})
        R, t = 0.0, 0.0
        if isCache and wc["grid"]:
            best = self._anchor(wc["grid"], T)
            if best is not None:
                self.snapshot_load(wc["grid"][best])
                R = float(wc["R"]) or 0.0
                t = best
        try:
            while t < T - 1e-9:
                R += 0.7071067811865476
                diag = False
                if R > 1:
                    R -= 1
                    diag = True
                dt_q = max(1, int(self._js_round(min(base_dt, T - t) / zs)))
                self._chk(self.lib.cg_etch_wet_step(self.h, int(matId), dt_q, int(diag)))
                t += dt_q * zs
                self._save_anchor(wc["grid"], t)
                wc["R"] = R
            self.identify_cavity()
        finally:
            self.lib.cg_release_scratch(self.h)

    def deposit_ALD(self, matId, totalThk, dt=None, isCache=False):
        """deposit_ALD(matId, totalThk, {dt, isCache}) (script_columngrid2.js:111-868)."""
        T = float(totalThk)
        T = 0.0 if T != T else max(0.0, T)
        if T <= 0:
            return
        inf = self.info()
        step = max(max(inf.dx, inf.dy, min(inf.dx, inf.dy)) if dt is None else float(dt), 1e-3)
        ac = self.sliderCache.setdefault("aldCache", {"exposedSegments": None, "gridWithCavity": {}})
        if not (isCache and ac["exposedSegments"] is not None):
            n = C.c_uint64()
            self._chk(self.lib.cg_ald_collect_seeds(self.h, T, C.byref(n)))
            ac["exposedSegments"] = n.value
        t_anchor = 0.0
        if isCache and ac["gridWithCavity"]:
            best = self._anchor(ac["gridWithCavity"], T)
            if best is not None:
                self.snapshot_load(ac["gridWithCavity"][best])
                t_anchor = best
        try:
            t = t_anchor + step
            while t < T - 1e-9:
                self._chk(self.lib.cg_ald_increment(self.h, int(matId), t))
                self._save_anchor(ac["gridWithCavity"], t)
                t += step
            self._chk(self.lib.cg_ald_coat(self.h, int(matId), T))
            self._save_anchor(ac["gridWithCavity"], T)
            self.identify_cavity()
        finally:
            self.lib.cg_release_scratch(self.h)

    def etch_wet_step(self, matId, dt_q, diagonal):
        self._chk(self.lib.cg_etch_wet_step(self.h, int(matId), int(dt_q), int(bool(diagonal))))

    def etch_wet_step_slab(self, matId, dt_q, diagonal, hmax_q, planes, ghost_lower, ghost_upper):
        """One etch_wet increment of a slab (no cavity pass): see cg_etch_wet_step_slab."""
        self._chk(self.lib.cg_etch_wet_step_slab(self.h, int(matId), int(dt_q), int(bool(diagonal)), int(hmax_q), int(planes),
                                                 ghost_lower or None, ghost_upper or None))

    # deposit_ALD of a slab: see cg_ald_collect_seeds_slab and the calls listed with it
    def ald_collect_seeds_slab(self, T, hmax_q, planes, ghost_lower, ghost_upper):
        """Seeds of this slab's rows; returns the most seeds held by one of its columns."""
        sp = C.c_int()
        self._chk(self.lib.cg_ald_collect_seeds_slab(self.h, float(T), int(hmax_q), int(planes), ghost_lower or None, ghost_upper or None,
                                                     C.byref(sp), None))
        return sp.value

    def ald_seed_halo_bytes(self, seed_planes):
        n = C.c_size_t()
        self._chk(self.lib.cg_ald_seed_halo_bytes(self.h, int(seed_planes), C.byref(n)))
        return n.value

    def ald_seed_halo_pack(self, seed_planes, send_lower, send_upper):
        self._chk(self.lib.cg_ald_seed_halo_pack(self.h, int(seed_planes), send_lower, send_upper))

    def ald_seed_halo_unpack(self, seed_planes, recv_lower, recv_upper):
        self._chk(self.lib.cg_ald_seed_halo_unpack(self.h, int(seed_planes), recv_lower or None, recv_upper or None))

    def ald_coat_slab(self, matId, t, hmax_q):
        self._chk(self.lib.cg_ald_coat_slab(self.h, int(matId), float(t), int(hmax_q)))

    def ald_merge_slab(self, hbase_q):
        self._chk(self.lib.cg_ald_merge_slab(self.h, int(hbase_q)))

    def ald_drop_seeds(self):
        self._chk(self.lib.cg_ald_drop_seeds(self.h))

    def release_scratch(self):
        self._chk(self.lib.cg_release_scratch(self.h))

    def set_plane_budget(self, max_planes):
        """Cap on the planes the handle maps before it compacts its columns instead (see cg_set_plane_budget)."""
        self._chk(self.lib.cg_set_plane_budget(self.h, int(max_planes)))

    def slider_sizes(self):
        """(ALD anchors, wet-etch anchors, seeds held) -- what the reference keeps in sliderCache."""
        ac = self.sliderCache.get("aldCache"); wc = self.sliderCache.get("wetetchCache")
        return (len(ac["gridWithCavity"]) if ac else 0, len(wc["grid"]) if wc else 0,
                (ac["exposedSegments"] or 0) if ac else 0)

    def addLayer(self, i, j, matId, z0_real, z1_real):
        """addLayer (script_columngrid.js:196-211) on one device column."""
        self._chk(self.lib.cg_add_layer(self.h, int(i), int(j), int(matId) & 255, float(z0_real), float(z1_real)))

    def removeTopLayer(self, i, j):
        """removeTopLayer (script_columngrid.js:213-218); the freed slot reads back as zeros."""
        self._chk(self.lib.cg_remove_top_layer(self.h, int(i), int(j)))

    def maxHeight(self):
        z = C.c_double()
        self._chk(self.lib.cg_max_height(self.h, C.byref(z), None))
        return z.value

    max_height = maxHeight

    def createNewGrid(self, LmaxInit=0):
        self._chk(self.lib.cg_reset(self.h, int(LmaxInit)))

    reset = createNewGrid

    def clearAll(self):
        self._chk(self.lib.cg_clear(self.h))

    # ---- split stencil steps (multi-GPU)
    def stencil_begin(self, kind, mask, matId, thickness, conformality, has_lower, has_upper):
        R = C.c_int()
        self._chk(self.lib.cg_stencil_begin(self.h, kind, self._mid(mask), int(matId), float(thickness),
                                            float(conformality), int(has_lower), int(has_upper), C.byref(R)))
        return R.value

    def halo_buffers(self):
        p = [C.c_void_p() for _ in range(4)]
        n = C.c_size_t()
        self._chk(self.lib.cg_halo_buffers(self.h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3]), C.byref(n)))
        return [q.value for q in p], n.value

    def stencil_finish(self):
        self._chk(self.lib.cg_stencil_finish(self.h))

    def stencil_interior(self):
        """The part of an open lateral step that needs no halo row (overlaps with the exchange)."""
        self._chk(self.lib.cg_stencil_interior(self.h))

    def set_halo_buffers(self, send_lower, send_upper, recv_lower, recv_upper, bytes_each):
        self._chk(self.lib.cg_set_halo_buffers(self.h, send_lower, send_upper, recv_lower, recv_upper, bytes_each))

    def profile(self, enable=True):
        """-> {kernel kind: (ms, launches)} accumulated since the previous call; sets the enabled state."""
        ms = (C.c_double * CG_NKIND)(); cnt = (C.c_uint64 * CG_NKIND)()
        self._chk(self.lib.cg_profile(self.h, 1 if enable else 0, ms, cnt))
        return {KERNEL_KINDS[k]: (ms[k], int(cnt[k])) for k in range(CG_NKIND) if cnt[k]}

    # ---- data exchange
    def download(self):
        """(len, mat, zpair) in the reference layout with stride Lmax."""
        inf = self.info()
        n, L = inf.nx * inf.ny, inf.lmax
        ln = np.empty(n, np.uint8); mat = np.empty(n * L, np.uint8); zp = np.empty(2 * n * L, np.uint16)
        self._chk(self.lib.cg_download(self.h, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data))
        return ln, mat, zp

    arrays = download

    @property
    def cols(self):
        ln, mat, zp = self.download()
        return {"mat": mat, "zpair": zp, "len": ln}

    def upload(self, ln, mat, zp, lmax):
        ln = np.ascontiguousarray(ln, np.uint8); mat = np.ascontiguousarray(mat, np.uint8)
        zp = np.ascontiguousarray(zp, np.uint16)
        n = self.NX * self.NY
        if ln.size != n or mat.size != n * lmax or zp.size != 2 * n * lmax:
            raise ValueError("array sizes do not match the grid")
        self._chk(self.lib.cg_upload(self.h, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data, int(lmax)))

    def snapshot_save(self, slot):
        self._chk(self.lib.cg_snapshot_save(self.h, int(slot)))

    def snapshot_load(self, slot):
        self._chk(self.lib.cg_snapshot_load(self.h, int(slot)))

    def snapshot_drop(self, slot):
        self._chk(self.lib.cg_snapshot_drop(self.h, int(slot)))

    def slices(self, axis, indices):
        """axis 'YZ' (fixed i) or 'XZ' (fixed j): list of (len, mat, zpair) per requested line."""
        ax = {"YZ": 0, "XZ": 1}[axis] if isinstance(axis, str) else int(axis)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        inf = self.info()
        line = inf.ny if ax == 0 else inf.nx
        L, cnt = inf.lmax, idx.size
        ln = np.empty(cnt * line, np.uint8); mat = np.empty(cnt * line * L, np.uint8)
        zp = np.empty(2 * cnt * line * L, np.uint16)
        self._chk(self.lib.cg_slices(self.h, ax, idx.ctypes.data, cnt, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data))
        return ln.reshape(cnt, line), mat.reshape(cnt, line, L), zp.reshape(cnt, line, L, 2)

    def slices_device(self, axis, indices):
        ax = {"YZ": 0, "XZ": 1}[axis] if isinstance(axis, str) else int(axis)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self._chk(self.lib.cg_slices_device(self.h, ax, idx.ctypes.data, idx.size, None, None, None))

    def slices_noalloc(self, axis, indices, d_mat, d_zpair, d_len):
        """Cross-sections into the caller's device buffers (pointers as ints) without any allocation: see cg_slices_noalloc."""
        ax = {"YZ": 0, "XZ": 1}[axis] if isinstance(axis, str) else int(axis)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self._chk(self.lib.cg_slices_noalloc(self.h, ax, idx.ctypes.data, idx.size, d_mat, d_zpair, d_len))

    def index_from_pos(self, axis, pos):
        out = C.c_int32()
        self._chk(self.lib.cg_index_from_pos(self.h, int(axis), float(pos), C.byref(out)))
        return out.value

    # ---- checks and measurement
    def checksum(self):
        s = C.c_uint64()
        self._chk(self.lib.cg_checksum(self.h, C.byref(s)))
        return s.value

    def fnv(self):
        """FNV-1a/32 of the live layers (host side, after a download); small grids only."""
        ln, mat, zp = self.download()
        L = self.Lmax
        h = 2166136261
        for c in range(ln.size):
            l = int(ln[c]); h = ((h ^ l) * 16777619) & 0xFFFFFFFF
            for k in range(l):
                b = c * L + k
                for v in (int(mat[b]), int(zp[2 * b]), int(zp[2 * b + 1])):
                    h = ((h ^ v) * 16777619) & 0xFFFFFFFF
        return "%08x" % h

    def stats_enable(self, on=True):
        self._chk(self.lib.cg_step_stats(self.h, 1 if on else 0, None))

    def stats(self):
        out = (C.c_uint64 * CG_NSTAT)()
        self._chk(self.lib.cg_step_stats(self.h, 1, out))
        return list(out)

    def timer_start(self):
        self._chk(self.lib.cg_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.cg_timer_stop(self.h, C.byref(ms)))
        return ms.value


class GroupMask:
    """A mask of a GroupGrid (one device raster per slab behind one id)."""

    def __init__(self, grid, mask_id):
        self.grid, self.id = grid, mask_id

    def download(self):
        out = np.empty(self.grid.NX * self.grid.NY, np.uint8)
        self.grid._chk(self.grid.lib.cg_group_mask_download(self.grid.h, self.id, out.ctypes.data, out.size))
        return out

    def destroy(self):
        if self.id is not None and self.grid.h:
            self.grid.lib.cg_group_mask_destroy(self.grid.h, self.id)
        self.id = None


class GroupGrid:
    """One grid cut into slabs that the LIBRARY drives (cg_group_*, include/colgrid.h): the reference's ColumnGrid call
    surface over several devices from one process.  devices: one HIP device ordinal per slab (repeats allowed)."""

    def __init__(self, LXnm, LYnm, dx=2.0, dy=2.0, LmaxInit=32, zScale=0.1, *, devices=(0,), lib=None):
        self.lib = lib or load_library()
        self.h = None
        nx = max(1, int(ColumnGrid._js_round(LXnm / dx)))                 # setDomain (script_columngrid.js:49-60)
        ny = max(1, int(ColumnGrid._js_round(LYnm / dy)))
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        out = C.c_void_p()
        rc = self.lib.cg_group_create(nx, ny, dx, dy, LmaxInit, zScale, devs, len(devices), C.byref(out))
        if rc:
            raise ColGridError(rc, (self.lib.cg_group_last_error(None) or b"").decode())
        self.h = out
        self.zScale = zScale

    def _chk(self, rc):
        if rc:
            raise ColGridError(rc, (self.lib.cg_group_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.lib.cg_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        inf = CgInfo()
        self._chk(self.lib.cg_group_get_info(self.h, C.byref(inf)))
        return inf

    NX = property(lambda s: s.info().nx)
    NY = property(lambda s: s.info().ny)
    Lmax = property(lambda s: s.info().lmax)

    def transport(self):
        """-> (uses RCCL, exchanges so far, bytes moved so far)."""
        r, e, b = C.c_int(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.cg_group_transport(self.h, C.byref(r), C.byref(e), C.byref(b)))
        return bool(r.value), e.value, b.value

    def sync(self):
        self._chk(self.lib.cg_group_sync(self.h))

    # ---- masks
    def mask_from_raster(self, raster):
        r = np.ascontiguousarray(raster, dtype=np.uint8).reshape(-1)
        mid = C.c_int()
        self._chk(self.lib.cg_group_mask_from_raster(self.h, r.ctypes.data, r.size, C.byref(mid)))
        return GroupMask(self, mid.value)

    def mask_from_shapes(self, reduced):
        arr, n, pts = _pack_shapes(reduced)
        mid = C.c_int()
        self._chk(self.lib.cg_group_mask_from_shapes(self.h, arr.ctypes.data if n else None, n,
                                                     pts.ctypes.data if pts.size else None, pts.size // 2, C.byref(mid)))
        return GroupMask(self, mid.value)

    @staticmethod
    def _mid(mask):
        if mask is None:
            return CG_MASK_NONE
        if isinstance(mask, GroupMask):
            return mask.id
        raise TypeError("mask must be None or a GroupMask made by this grid")

    # ---- the reference's step methods
    def deposit_general(self, mask, matId, thickness, conformality=1.0):
        self._chk(self.lib.cg_group_deposit_general(self.h, self._mid(mask), int(matId), float(thickness), float(conformality)))

    def etch_general(self, mask, matId, thickness, conformality=1.0):
        self._chk(self.lib.cg_group_etch_general(self.h, self._mid(mask), int(matId), float(thickness), float(conformality)))

    def cmp(self, depth, stopMatId=0):
        self._chk(self.lib.cg_group_cmp(self.h, float(depth), int(stopMatId)))

    def strip_connected(self, matId):
        self._chk(self.lib.cg_group_strip_connected(self.h, int(matId)))

    def identify_cavity(self):
        self._chk(self.lib.cg_group_identify_cavity(self.h))

    def etch_wet(self, matId, totalThk, dt=None):
        self._chk(self.lib.cg_group_etch_wet(self.h, int(matId), float(totalThk), float("nan") if dt is None else float(dt)))

    def deposit_ALD(self, matId, totalThk, dt=None):
        self._chk(self.lib.cg_group_deposit_ald(self.h, int(matId), float(totalThk), float("nan") if dt is None else float(dt)))

    def createNewGrid(self, LmaxInit=0):
        self._chk(self.lib.cg_group_reset(self.h, int(LmaxInit)))

    def maxHeight(self):
        z = C.c_double()
        self._chk(self.lib.cg_group_max_height(self.h, C.byref(z), None))
        return z.value

    def checksum(self):
        s = C.c_uint64()
        self._chk(self.lib.cg_group_checksum(self.h, C.byref(s)))
        return s.value

    def download(self):
        """-> (len, mat, zpair) of the whole grid in the reference layout."""
        inf = self.info()
        n, L = inf.nx * inf.ny, inf.lmax
        ln = np.empty(n, np.uint8); mat = np.empty(n * L, np.uint8); zp = np.empty(2 * n * L, np.uint16)
        self._chk(self.lib.cg_group_download(self.h, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data))
        return ln, mat, zp

    def upload(self, ln, mat, zp, lmax):
        ln = np.ascontiguousarray(ln, np.uint8); mat = np.ascontiguousarray(mat, np.uint8); zp = np.ascontiguousarray(zp, np.uint16)
        self._chk(self.lib.cg_group_upload(self.h, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data, int(lmax)))

    def snapshot_save(self, slot):
        self._chk(self.lib.cg_group_snapshot_save(self.h, int(slot)))

    def snapshot_load(self, slot):
        self._chk(self.lib.cg_group_snapshot_load(self.h, int(slot)))

    def snapshot_drop(self, slot):
        self._chk(self.lib.cg_group_snapshot_drop(self.h, int(slot)))

    def slices(self, axis, indices):
        ax = {"YZ": 0, "XZ": 1}[axis] if isinstance(axis, str) else int(axis)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        inf = self.info()
        line = inf.ny if ax == 0 else inf.nx
        L, cnt = inf.lmax, idx.size
        ln = np.empty(cnt * line, np.uint8); mat = np.empty(cnt * line * L, np.uint8)
        zp = np.empty(2 * cnt * line * L, np.uint16)
        self._chk(self.lib.cg_group_slices(self.h, ax, idx.ctypes.data, cnt, mat.ctypes.data, zp.ctypes.data, ln.ctypes.data))
        return ln.reshape(cnt, line), mat.reshape(cnt, line, L), zp.reshape(cnt, line, L, 2)
