"""Loading of the fixtures under tests/golden/ (outputs of the reference, see oracle/gen_golden.js)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STEP_CASES = ["vertical_small", "stencil_small", "stencil_odd", "anchor_256", "masks_small",
              "growth_small", "cavity_small", "empty_mask_small", "kcache_small", "wet_ald_small", "wet_ald_cache", "wet_ald_thin", "accessors_small",
              "unequal_pitch", "unequal_pitch_t"]


def load_case(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        case = json.load(f)
    n = case["nx"] * case["ny"]
    case["dump_arrays"] = []
    p = os.path.join(GOLDEN, name + ".bin.gz")
    if os.path.exists(p) and case.get("dumps"):
        raw = np.frombuffer(gzip.open(p, "rb").read(), dtype=np.uint8)
        off = 0
        for d in case["dumps"]:
            lm = d["lmax"]
            ln = raw[off:off + n]; off += n
            mat = raw[off:off + n * lm]; off += n * lm
            zp = raw[off:off + 4 * n * lm].view(np.uint16); off += 4 * n * lm
            case["dump_arrays"].append((ln, mat, zp, lm))
        assert off == raw.size
    case["rasters"] = {}
    p = os.path.join(GOLDEN, name + ".masks.gz")
    if os.path.exists(p):
        raw = np.frombuffer(gzip.open(p, "rb").read(), dtype=np.uint8)
        for k, m in enumerate(case["mask_order"]):
            case["rasters"][m] = raw[k * n:(k + 1) * n]
    return case


def fnv_bytes(a):
    h = 2166136261
    for b in np.asarray(a, dtype=np.uint8).reshape(-1).tolist():
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return "%08x" % h


def run_steps(grid, case, mask_of, on_step=None):
    """Drive `grid` (oracle or engine wrapper with the reference's method names) through the
    steps of a golden case.  mask_of(name) -> whatever the grid takes as a mask."""
    for idx, s in enumerate(case["steps"]):
        op = s["op"]
        if op in ("deposit_general", "etch_general"):
            m = s.get("mask")
            mask = None if m is None else mask_of(m)
            getattr(grid, op)(mask, s["mat"], s["t"], s["S"])
        elif op == "cmp":
            grid.cmp(s["t"], s["mat"])
        elif op == "strip_connected":
            grid.strip_connected(s["mat"])
        elif op == "identify_cavity":
            grid.identify_cavity()
        elif op == "reset":
            grid.reset()
        elif op == "etch_wet":
            grid.etch_wet(s["mat"], s["t"], dt=s.get("dt"), isCache=s.get("isCache", False))
        elif op == "deposit_ALD":
            grid.deposit_ALD(s["mat"], s["t"], dt=s.get("dt"), isCache=s.get("isCache", False))
        elif op == "slider_reset":
            grid.slider_reset()
        elif op == "addLayer":
            grid.addLayer(s["i"], s["j"], s["mat"], s["z0"], s["z1"])
        elif op == "removeTopLayer":
            grid.removeTopLayer(s["i"], s["j"])
        else:
            raise ValueError(op)
        if on_step:
            on_step(idx, s)
