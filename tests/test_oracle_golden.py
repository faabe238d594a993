"""The CPU restatement (oracle/colgrid_oracle.c) against outputs of the reference itself.

Every fixture under tests/golden/ was produced by the reference's JavaScript running in Node
(oracle/gen_golden.js).  Equality is required on the complete len / mat / zpair arrays, dead
slots included, after every dumped step, and on the FNV hash of the live layers after every step.
"""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, STEP_CASES, load_case, fnv_bytes, run_steps
from oracle import OracleGrid


def _grid_for(case):
    d = case["domain"]
    return OracleGrid(d["LX"], d["LY"], d["dx"], d["dy"], d.get("Lmax", 32), d.get("zScale", 0.1))


@pytest.mark.parametrize("name", STEP_CASES)
def test_steps_match_reference(name):
    case = load_case(name)
    g = _grid_for(case)
    assert (g.NX, g.NY) == (case["nx"], case["ny"])
    n = g.NX * g.NY
    rasters = {}

    def mask_of(m):
        if m == "all":
            return np.ones(n, np.uint8)
        if m not in rasters:
            rasters[m] = g.mask_raster(case["reduced"][m])
        return rasters[m]

    def check(idx, s):
        assert g.fnv() == s["fnv"], f"{name} step {idx} {s['op']}"
        assert g.Lmax == s["lmax"]
        assert g.max_height() == s["max_height"]
        if "slider" in s:
            na, nw, nseed = g.slider_sizes()
            assert (na, nw, nseed) == (len(s["slider"]["ald_keys"]), len(s["slider"]["wet_keys"]), s["slider"]["seeds"])
        if "probe" in s:                                  # getLen / topMat / topZ of the edited column
            oln, omat, ozp = g.arrays()
            c = s["i"] * g.NY + s["j"]; l = int(oln[c])
            top = (c * g.Lmax + l - 1)
            got = [l, int(omat[top]) if l else 0, float(ozp[2 * top + 1]) * case["domain"].get("zScale", 0.1) if l else 0]
            assert got == s["probe"], f"{name} step {idx} probe"
        if "dump_index" in s:
            ln, mat, zp, lm = case["dump_arrays"][s["dump_index"]]
            oln, omat, ozp = g.arrays()
            assert lm == g.Lmax
            np.testing.assert_array_equal(oln, ln, err_msg=f"{name} step {idx} len")
            np.testing.assert_array_equal(omat, mat, err_msg=f"{name} step {idx} mat")
            np.testing.assert_array_equal(ozp, zp, err_msg=f"{name} step {idx} zpair")

    run_steps(g, case, mask_of, check)


@pytest.mark.parametrize("name", [c for c in STEP_CASES if c != "anchor_256"])
def test_mask_rasters_match_reference(name):
    case = load_case(name)
    g = _grid_for(case)
    for m in case["mask_order"]:
        got = g.mask_raster(case["reduced"][m])
        np.testing.assert_array_equal(got, case["rasters"][m], err_msg=f"{name} mask {m}")


def test_anchor_raster_hash():
    case = load_case("anchor_256")
    g = _grid_for(case)
    for m, want in zip(case["mask_order"], case["raster_fnv"]):
        assert fnv_bytes(g.mask_raster(case["reduced"][m])) == want


@pytest.mark.parametrize("nx", [64, 256])
def test_runtime_c1(nx):
    """Config C1 (SUBSTR -> DEPO -> masked ETCH) as produced by the reference RUNTIME."""
    with open(os.path.join(GOLDEN, f"runtime_c1_{nx}.json")) as f:
        c = json.load(f)
    d = c["domain"]
    g = OracleGrid(d["LX"], d["LY"], d["dx"], d["dy"])
    raster = g.mask_raster(c["reduced"])
    assert int(raster.sum()) == c["open_cells"]
    assert fnv_bytes(raster) == c["raster_fnv"]
    n = g.NX * g.NY
    ids = {k: v["id"] for k, v in c["materialColor"].items()}
    for p in c["processes"]:
        mask = np.ones(n, np.uint8) if p["mask"] == "-" else raster
        S = p.get("conformality", 0) if p["kind"] != "SUBSTR" else 0
        if p["kind"] in ("SUBSTR", "DEPO"):
            g.deposit_general(mask, ids[p["material"]], p["thickness"], S)
        else:
            g.etch_general(mask, ids[p["material"]], p["thickness"], S)
    assert g.fnv() == c["fnv"]
    ln = g.arrays()[0]
    hist = {str(k): int(v) for k, v in zip(*np.unique(ln, return_counts=True))}
    assert hist == c["len_hist"]


def test_kernel_taps():
    with open(os.path.join(GOLDEN, "kernel_taps.json")) as f:
        ks = json.load(f)
    for k in ks:
        g = OracleGrid(8 * k["dx"], 8 * k["dy"], k["dx"], k["dy"])
        Rx, du, dv, add = g.kernel_taps(k["t"], k["S"])
        assert Rx == k["Rx"]
        want = np.array(k["taps"], dtype=np.int64).reshape(-1, 3)
        np.testing.assert_array_equal(du, want[:, 0])
        np.testing.assert_array_equal(dv, want[:, 1])
        np.testing.assert_array_equal(add.view(np.uint32), want[:, 2].astype(np.uint32))


def test_quantisation_known_answers():
    """round(z/0.1) with halves toward +inf and the Uint16 wrap, via a one-column deposit."""
    with open(os.path.join(GOLDEN, "quantize.json")) as f:
        q = json.load(f)
    for rec in q["quant"]:
        if rec["z"] <= 0:
            continue
        g = OracleGrid(2, 2, 2, 2)
        g.deposit_general(None, 3, rec["z"], 0)
        ln, mat, zp = g.arrays()
        assert ln[0] == 1 and zp[0] == 0 and zp[1] == rec["u16"], rec
