"""HIP engine (through the C ABI) against the reference's golden outputs and the CPU oracle.

Bit-exact bar: len / mat / zpair arrays must be identical, dead slots included.
"""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, STEP_CASES, load_case, fnv_bytes, run_steps
from oracle import OracleGrid

pytestmark = pytest.mark.gpu


def _engine():
    import fem_amd
    return fem_amd


def _grid_for(case):
    d = case["domain"]
    return _engine().ColumnGrid(d["LX"], d["LY"], d["dx"], d["dy"], d.get("Lmax", 32), d.get("zScale", 0.1))


def _assert_same(g, ref_arrays, what):
    ln, mat, zp = g.download()
    rl, rm, rz = ref_arrays
    np.testing.assert_array_equal(ln, rl, err_msg=what + " len")
    np.testing.assert_array_equal(mat, rm, err_msg=what + " mat")
    np.testing.assert_array_equal(zp, rz, err_msg=what + " zpair")


@pytest.mark.parametrize("name", STEP_CASES)
def test_golden_steps(name):
    case = load_case(name)
    g = _grid_for(case)
    assert (g.NX, g.NY) == (case["nx"], case["ny"])
    n = g.NX * g.NY
    masks = {}

    def mask_of(m):
        if m not in masks:
            masks[m] = g.mask_from_raster(np.ones(n, np.uint8)) if m == "all" else g.mask_from_shapes(case["reduced"][m])
        return masks[m]

    small = n <= 16384

    def check(idx, s):
        assert g.Lmax == s["lmax"], f"{name} step {idx}"
        assert g.maxHeight() == s["max_height"], f"{name} step {idx}"
        if "slider" in s:
            assert g.slider_sizes() == (len(s["slider"]["ald_keys"]), len(s["slider"]["wet_keys"]), s["slider"]["seeds"])
        if "probe" in s:                                  # getLen / topMat / topZ of the edited column
            ln, mat, zp = g.download()
            c = s["i"] * g.NY + s["j"]; l = int(ln[c]); top = c * g.Lmax + l - 1
            got = [l, int(mat[top]) if l else 0, float(zp[2 * top + 1]) * case["domain"].get("zScale", 0.1) if l else 0]
            assert got == s["probe"], f"{name} step {idx} probe"
        if "dump_index" in s:
            ln, mat, zp, lm = case["dump_arrays"][s["dump_index"]]
            _assert_same(g, (ln, mat, zp), f"{name} step {idx} {s['op']}")
        if small or idx == len(case["steps"]) - 1:
            assert g.fnv() == s["fnv"], f"{name} step {idx} {s['op']}"

    run_steps(g, case, mask_of, check)
    g.close()


@pytest.mark.parametrize("name", [c for c in STEP_CASES if c != "anchor_256"])
def test_golden_mask_rasters(name):
    case = load_case(name)
    g = _grid_for(case)
    for m in case["mask_order"]:
        got = g.mask_from_shapes(case["reduced"][m]).download()
        np.testing.assert_array_equal(got, case["rasters"][m], err_msg=f"{name} mask {m}")
    g.close()


@pytest.mark.parametrize("nx", [64, 256])
def test_runtime_c1_golden(nx):
    with open(os.path.join(GOLDEN, f"runtime_c1_{nx}.json")) as f:
        c = json.load(f)
    d = c["domain"]
    g = _engine().ColumnGrid(d["LX"], d["LY"], d["dx"], d["dy"])
    m = g.mask_from_shapes(c["reduced"])
    raster = m.download()
    assert int(raster.sum()) == c["open_cells"] and fnv_bytes(raster) == c["raster_fnv"]
    ids = {k: v["id"] for k, v in c["materialColor"].items()}
    for p in c["processes"]:
        mask = None if p["mask"] == "-" else m
        S = p.get("conformality", 0) if p["kind"] != "SUBSTR" else 0
        if p["kind"] in ("SUBSTR", "DEPO"):
            g.deposit_general(mask, ids[p["material"]], p["thickness"], S)
        else:
            g.etch_general(mask, ids[p["material"]], p["thickness"], S)
    assert g.fnv() == c["fnv"]
    g.close()


def _lcg(seed):
    s = seed
    while True:
        s = (s * 1103515245 + 12345) & 0x7FFFFFFF
        yield s


def _random_mask_data(rng, L, nshapes):
    objs = []
    for _ in range(nshapes):
        kind = next(rng) % 4
        left, top = next(rng) % L, next(rng) % L
        w, h = 4 + next(rng) % (L // 3), 4 + next(rng) % (L // 3)
        ang = [0, 0, 15, 30, -40, 77.5][next(rng) % 6]
        pol = "negative" if next(rng) % 3 == 0 else "positive"
        o = {"left": left, "top": top, "width": w, "height": h, "angle": ang, "scaleX": 1, "scaleY": 1,
             "visible": True, "data": {"polarity": pol}}
        if kind == 0:
            o["type"] = "rect"
        elif kind == 1:
            o.update(type="circle", radius=w / 2, height=w)
        elif kind == 2:
            o.update(type="ellipse", rx=w / 2, ry=h / 2)
        else:
            o.update(type="polygon", points=[{"x": 0, "y": 0}, {"x": w, "y": h / 3}, {"x": w / 2, "y": h}, {"x": -w / 4, "y": h / 2}])
        objs.append(o)
    return {"objects": objs}


@pytest.mark.parametrize("nshapes,slab", [(7, None), (300, None), (300, (70, 150)), (2500, None)])
def test_mask_rasteriser_many_shapes(nshapes, slab):
    """_isPointBlocked (script_runtime.js:54-141) with the shapes binned by tile (8 shapes or more) and walked
    one by one (fewer): rectangles, circles, ellipses and polygons, rotated, scaled and mirrored, of both
    polarities, some far larger than a tile, some off the grid; on a whole grid and on a slab of it."""
    fem = _engine()
    rng = _lcg(1000 + nshapes)
    nx, ny = 290, 333
    data = _random_mask_data(rng, 2 * max(nx, ny), nshapes)
    for k, o in enumerate(data["objects"]):
        if k % 5 == 1:
            o["scaleX"], o["scaleY"] = [(0.4, 1.7), (-1.0, 1.0), (2.5, -0.6)][k % 3]
        if k % 11 == 3:
            o["left"] -= 3 * nx; o["width"] *= 6; o["height"] *= 5        # large, partly off the grid
        if k % 17 == 4:
            o["left"] += 5 * nx                                           # entirely off the grid
        if nshapes > 1000:
            o["width"] = 2 + o["width"] / 12; o["height"] = 2 + o["height"] / 12
            if o["type"] == "circle":
                o["radius"] = o["width"] / 2; o["height"] = o["width"]
            if o["type"] == "ellipse":
                o["rx"], o["ry"] = o["width"] / 2, o["height"] / 2
    red = fem.reduce_mask(data)
    want = OracleGrid(2 * nx, 2 * ny, 2, 2).mask_raster(red).reshape(nx, ny)
    if slab is None:
        g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    else:
        g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(nx, ny, slab[0], slab[1]))
        want = want[slab[0]:slab[0] + slab[1]]
    got = g.mask_from_shapes(red).download()
    assert 0 < int(want.sum()) < want.size
    np.testing.assert_array_equal(got, want.reshape(-1))
    g.close()


def test_mask_rasteriser_periodic_at_scale():
    """The benchmark's periodic masks as shape lists (16 384 squares, 29 584 discs on 4096 x 4096 columns)
    against their closed form on the cell centres."""
    fem = _engine()
    from fem_amd import flows
    n = 4096
    specs = flows.mask_specs(2.0 * n)
    g = fem.ColumnGrid(2.0 * n, 2.0 * n, 2, 2)
    for name in ("M", "Mshift", "Mc"):
        red = fem.reduce_mask(flows.shapes_of(specs[name], 2 * n, 2 * n))
        m = g.mask_from_shapes(red)
        np.testing.assert_array_equal(m.download(), flows.raster_numpy(specs[name], n, n), err_msg=name)
        m.destroy()
    g.close()


@pytest.mark.parametrize("stats", [True, False])
@pytest.mark.parametrize("nx,ny,seed", [(192, 320, 1), (333, 257, 2), (512, 512, 3), (100, 90, 4)])   # the last: 4-column vertical kernels, one-column lateral ones
def test_random_flow_vs_oracle(nx, ny, seed, stats):
    """A seeded multi-step flow with every supported step kind on grids the oracle does in seconds.  With statistics
    the engine runs the kernel instantiations that count cases (and stay on the reference's double arithmetic
    throughout); without, the ones a production run uses (row-streaming lateral kernel, integer vertical steps)."""
    fem = _engine()
    rng = _lcg(seed)
    L = 2 * max(nx, ny)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    masks = []
    for _ in range(3):
        red = fem.reduce_mask(_random_mask_data(rng, L, 12))
        gm = g.mask_from_shapes(red)
        om = o.mask_raster(red)
        np.testing.assert_array_equal(gm.download(), om)
        masks.append((gm, om))
    g.stats_enable(stats)
    steps = [("d", None, 3, 30, 0), ("d", None, 4, 20, 0)]
    for k in range(14):
        kind = "d" if next(rng) % 2 == 0 else "e"
        mi = next(rng) % 4
        mat = [3, 4, 5, 6, 255][next(rng) % (5 if kind == "e" else 4)]
        t = [2.5, 4, 6, 7.3, 10, 13.05, 21][next(rng) % 7]
        S = [0, 0, 0.5, 0.85, 1.0][next(rng) % 5]
        steps.append((kind, mi if mi < 3 else None, mat, t, S))
    for idx, (kind, mi, mat, t, S) in enumerate(steps):
        gm, om = (None, None) if mi is None else masks[mi]
        if kind == "d":
            g.deposit_general(gm, mat, t, S); o.deposit_general(om, mat, t, S)
        else:
            g.etch_general(gm, mat, t, S); o.etch_general(om, mat, t, S)
        assert g.checksum() == o.checksum(), f"step {idx} {kind} mat={mat} t={t} S={S} mask={mi}"
        if stats:
            gs, os_ = g.stats(), o.stats()
            assert gs[2:6] == os_[2:6], f"step {idx} stats {gs} vs {os_}"
    assert g.Lmax == o.Lmax
    _assert_same(g, o.arrays(), "final")
    g.close()


@pytest.mark.parametrize("zero_tops", [False, True])
def test_cavity_pass_next_to_empty_columns_and_columns_at_height_zero(zero_tops):
    """identify_cavity (cg2:9-83) on a grid with empty columns (a gap next to one is exposed, cg1:897) and, in the second
    case, with columns whose only layer ends at height 0: the cavity layer the reference lays on such a column rests on the
    floor and is NOT exposed (cg1:886), so it stays unless the flood reaches it -- the engine's pass that leaves those
    layers out steps aside for the literal one there.  Both against the oracle, arrays and all."""
    fem = _engine()
    rng = _lcg(91)
    nx, ny = 96, 128
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    masks = []
    for _ in range(3):
        red = fem.reduce_mask(_random_mask_data(rng, 2 * max(nx, ny), 6))
        masks.append((g.mask_from_shapes(red), o.mask_raster(red)))
    def both(fn, mi, *a):
        if fn in ("strip_connected", "identify_cavity"):
            getattr(g, fn)(*a); getattr(o, fn)(*a)
        else:
            gm, om = (None, None) if mi is None else masks[mi]
            getattr(g, fn)(gm, *a); getattr(o, fn)(om, *a)
        assert g.checksum() == o.checksum(), f"{fn} {mi} {a}"
        assert g.maxHeight() == o.max_height() and g.Lmax == o.Lmax, f"{fn} {mi} {a}"
    if zero_tops:
        both("deposit_general", 0, 3, 0.04, 0)             # rounds to a layer from 0 to 0 where mask 0 is open
    for mat, t in ((4, 10), (5, 6), (6, 8)):
        both("deposit_general", 1, mat, t, 0)
    both("etch_general", 2, 255, 30, 0)                    # open the stacks of mask 1 where mask 2 is open too
    both("strip_connected", None, 5)                       # the exposed pieces of material 5 go: gaps under material 6
    both("identify_cavity", None)
    both("deposit_general", 1, 7, 4, 1.0)                  # a lateral deposit runs the pass again
    both("strip_connected", None, 4)
    both("identify_cavity", None)
    both("etch_general", 1, 255, 5, 1.0)
    both("identify_cavity", None)
    # material 2 laid down as a film (vertical and conformal): its layers are nodes of the pass like any cavity layer --
    # the ones on the highest columns are exposed, the others are reached through the air above lower neighbours
    both("deposit_general", 0, 2, 3, 0)
    both("identify_cavity", None)
    both("deposit_general", 2, 2, 5, 1.0)
    both("deposit_general", None, 8, 2, 0)
    both("identify_cavity", None)
    both("etch_general", 2, 8, 3, 0)
    both("identify_cavity", None)
    _assert_same(g, o.arrays(), "final")
    g.close()


@pytest.mark.parametrize("nx,ny,seed", [(160, 224, 11), (301, 128, 12), (100, 90, 13)])
def test_random_flow_with_strip_cmp_cavity_vs_oracle(nx, ny, seed):
    """Flows that create buried gaps and enclosed cavities: strip_connected, identify_cavity (through
    conformal deposits and cmp) and cmp with and without a stopper, against the oracle step by step."""
    fem = _engine()
    rng = _lcg(seed)
    L = 2 * max(nx, ny)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    masks = []
    for _ in range(3):
        red = fem.reduce_mask(_random_mask_data(rng, L, 10))
        masks.append((g.mask_from_shapes(red), o.mask_raster(red)))
    steps = [("d", None, 3, 20, 0), ("d", None, 4, 8, 0), ("d", None, 5, 12, 0), ("e", 0, 255, 25, 0),
             ("s", None, 4, 0, 0), ("c", None, 0, 0, 0), ("d", None, 6, 6, 1.0), ("e", 1, 255, 30, 0),
             ("s", None, 4, 0, 0), ("d", None, 7, 5, 1.0), ("e", 2, 7, 4, 1.0), ("d", 1, 3, 9, 0.6),
             ("p", None, 0, 4, 0), ("s", None, 6, 0, 0), ("d", None, 4, 7, 1.0), ("p", None, 5, 30, 0),
             ("e", None, 255, 6, 1.0), ("p", None, 3, 40, 0), ("s", None, 3, 0, 0), ("c", None, 0, 0, 0)]
    for idx, (kind, mi, mat, t, S) in enumerate(steps):
        gm, om = (None, None) if mi is None else masks[mi]
        if kind == "d":
            g.deposit_general(gm, mat, t, S); o.deposit_general(om, mat, t, S)
        elif kind == "e":
            g.etch_general(gm, mat, t, S); o.etch_general(om, mat, t, S)
        elif kind == "s":
            g.strip_connected(mat); o.strip_connected(mat)
        elif kind == "c":
            g.identify_cavity(); o.identify_cavity()
        elif kind == "p":
            g.cmp(t, mat); o.cmp(t, mat)
        assert g.checksum() == o.checksum(), f"step {idx} {kind} mat={mat} t={t} S={S} mask={mi}"
        assert g.maxHeight() == o.max_height()
    ln, mat, zp = g.download()
    oln, omat, ozp = o.arrays()
    L0 = min(g.Lmax, o.Lmax)
    np.testing.assert_array_equal(ln, oln)
    np.testing.assert_array_equal(mat.reshape(-1, g.Lmax)[:, :L0], omat.reshape(-1, o.Lmax)[:, :L0])
    np.testing.assert_array_equal(zp.reshape(-1, g.Lmax, 2)[:, :L0], ozp.reshape(-1, o.Lmax, 2)[:, :L0])
    # upload of a state with gaps and cavities, then keep going on both sides
    g2 = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    g2.upload(oln, omat, ozp, o.Lmax)
    assert g2.checksum() == o.checksum()
    g2.etch_general(None, 255, 7, 0); o.etch_general(None, 255, 7, 0)
    g2.deposit_general(None, 8, 5, 1.0); o.deposit_general(None, 8, 5, 1.0)
    assert g2.checksum() == o.checksum()
    g.close(); g2.close()


@pytest.mark.parametrize("nx,ny,seed", [(96, 128, 21), (150, 77, 22), (168, 116, 34)])      # (seed 34: a state on which two cavity passes in a row differ from one)
def test_random_flow_with_wet_etch_and_ald_vs_oracle(nx, ny, seed):
    """etch_wet and deposit_ALD mixed with the other step kinds, against the oracle step by step:
    undercuts, pinched-off cavities, thin layers, every-solid targets, non-default increments."""
    fem = _engine()
    rng = _lcg(seed)
    L = 2 * max(nx, ny)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    g.keep_step_snapshots = False
    masks = []
    for _ in range(3):
        red = fem.reduce_mask(_random_mask_data(rng, L, 10))
        masks.append((g.mask_from_shapes(red), o.mask_raster(red)))
    steps = [("d", None, 3, 20, 0), ("d", None, 4, 6, 0), ("d", None, 5, 1.3, 0), ("d", None, 4, 0.7, 0), ("d", None, 5, 12, 0),
             ("e", 0, 255, 25, 0), ("w", None, 4, 3, None), ("a", None, 6, 3, None), ("e", 1, 255, 18, 0),
             ("w", None, 255, 2.5, None), ("a", None, 7, 5, None), ("d", None, 3, 4, 1.0), ("w", None, 3, 4.2, 1.5),
             ("e", 2, 255, 40, 0), ("a", None, 4, 2.7, 0.9), ("p", None, 0, 5, 0), ("w", None, 7, 1.1, 0.3),
             ("a", None, 5, 0.35, None), ("s", None, 4, 0, 0), ("a", None, 8, 6.5, 2.5), ("w", None, 255, 3, None)]
    for idx, (kind, mi, mat, t, S) in enumerate(steps):
        gm, om = (None, None) if mi is None else masks[mi]
        if kind == "d":
            g.deposit_general(gm, mat, t, S); o.deposit_general(om, mat, t, S)
        elif kind == "e":
            g.etch_general(gm, mat, t, S); o.etch_general(om, mat, t, S)
        elif kind == "s":
            g.strip_connected(mat); o.strip_connected(mat)
        elif kind == "p":
            g.cmp(t, mat); o.cmp(t, mat)
        elif kind == "w":
            g.etch_wet(mat, t, dt=S); o.etch_wet(mat, t, dt=S)
        elif kind == "a":
            g.deposit_ALD(mat, t, dt=S); o.deposit_ALD(mat, t, dt=S)
            assert g.slider_sizes()[2] == o.slider_sizes()[2], f"step {idx}: seeds"
        assert g.checksum() == o.checksum(), f"step {idx} {kind} mat={mat} t={t} S={S} mask={mi}"
        assert g.maxHeight() == o.max_height()
        assert o.arrays()[0].max() < o.Lmax, "the reference loses its state when a column fills every slot"
    ln, mat, zp = g.download()
    oln, omat, ozp = o.arrays()
    L0 = min(g.Lmax, o.Lmax)
    np.testing.assert_array_equal(ln, oln)
    np.testing.assert_array_equal(mat.reshape(-1, g.Lmax)[:, :L0], omat.reshape(-1, o.Lmax)[:, :L0])
    np.testing.assert_array_equal(zp.reshape(-1, g.Lmax, 2)[:, :L0], ozp.reshape(-1, o.Lmax, 2)[:, :L0])
    g.close()


def test_wet_etch_and_ald_whole_run_entry_points():
    """cg_etch_wet / cg_deposit_ald (the loops inside the library) equal the oracle's uncached runs."""
    fem = _engine()
    g = fem.ColumnGrid(128, 96, 2, 2)
    o = OracleGrid(128, 96, 2, 2)
    red = fem.reduce_mask({"objects": [{"type": "rect", "left": 0, "top": 0, "width": 400, "height": 400, "visible": True, "data": {"polarity": "positive"}},
                                       {"type": "rect", "left": 30, "top": 10, "width": 12, "height": 60, "visible": True, "data": {"polarity": "negative"}},
                                       {"type": "circle", "left": 70, "top": 30, "width": 16, "height": 16, "radius": 8, "visible": True, "data": {"polarity": "negative"}}]})
    gm, om = g.mask_from_shapes(red), o.mask_raster(red)
    for mat, t in ((3, 20), (4, 7), (5, 11)):
        g.deposit_general(None, mat, t, 0); o.deposit_general(None, mat, t, 0)
    g.etch_general(gm, 255, 24, 0); o.etch_general(om, 255, 24, 0)
    nan = float("nan")
    for fn, args in (("cg_etch_wet", (4, 5.0, nan)), ("cg_deposit_ald", (6, 4.0, nan)), ("cg_etch_wet", (255, 2.2, 0.7)),
                     ("cg_deposit_ald", (7, 3.3, 1.0)), ("cg_etch_wet", (6, 0.0, nan)), ("cg_deposit_ald", (7, -1.0, nan))):
        g._chk(getattr(g.lib, fn)(g.h, *args))
        dt = None if args[2] != args[2] else args[2]
        (o.etch_wet if fn == "cg_etch_wet" else o.deposit_ALD)(args[0], args[1], dt=dt)
        o.slider_reset()
        assert g.checksum() == o.checksum(), (fn, args)
    _assert_same(g, o.arrays(), "whole-run entry points")
    g.close()


def test_same_material_runs_across_gaps_and_form_changes():
    """Isotropic etch whose floor lies under a gap (two layers of one material separated by a stripped
    layer), appends and removals right after the kernels that restructure columns, and snapshots taken
    in either internal form of the columns -- against the oracle after every step."""
    fem = _engine()
    nx, ny = 120, 88
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    rng = _lcg(5)
    masks = []
    for _ in range(2):
        red = fem.reduce_mask(_random_mask_data(rng, 2 * nx, 8))
        masks.append((g.mask_from_shapes(red), o.mask_raster(red)))
    snap = {}

    def both(fn):
        fn(g, 0); fn(o, 1)

    def step(kind, mi=None, mat=0, t=0.0, S=0.0):
        gm, om = (None, None) if mi is None else masks[mi]
        for grid, m in ((g, gm), (o, om)):
            if kind == "d":
                grid.deposit_general(m, mat, t, S)
            elif kind == "e":
                grid.etch_general(m, mat, t, S)
            elif kind == "s":
                grid.strip_connected(mat)
            elif kind == "p":
                grid.cmp(t, mat)
            elif kind == "c":
                grid.identify_cavity()
        assert g.checksum() == o.checksum(), (kind, mi, mat, t, S)

    for args in (("d", None, 3, 20), ("d", None, 4, 5), ("d", None, 3, 6), ("d", None, 4, 3), ("d", None, 3, 4),
                 ("e", 0, 255, 16), ("s", None, 4)):
        step(*args)
    g.snapshot_save(1); snap[1] = o.arrays() + (o.Lmax,)           # columns are whole stacks here
    for args in (("e", None, 3, 9, 1.0), ("e", 1, 3, 2.5, 0.6), ("d", None, 3, 2, 0), ("d", None, 5, 3, 1.0),
                 ("e", None, 5, 3, 0), ("e", None, 3, 30, 1.0)):
        step(*args)
    g.snapshot_save(2); snap[2] = o.arrays() + (o.Lmax,)           # split form here
    for args in (("p", None, 0, 3), ("d", 1, 3, 7, 0.8), ("s", None, 3), ("d", None, 6, 4, 0), ("e", 0, 6, 4, 0), ("c",)):
        step(*args)
    for slot in (1, 2):
        g.snapshot_load(slot); ln, mat, zp, lm = snap[slot]; o.upload(ln, mat, zp, lm)
        assert g.checksum() == o.checksum()
        for args in (("e", None, 3, 5, 1.0), ("d", None, 3, 2.2, 0), ("e", 1, 255, 11, 0), ("d", None, 7, 6, 1.0)):
            step(*args)
    assert g.Lmax == o.Lmax
    _assert_same(g, o.arrays(), "final")
    g.close()


def test_lmax_follows_the_reference_while_columns_have_gaps():
    """A stripped layer leaves a gap that takes an interval on the device but no slot in the
    reference's arrays: the Lmax the engine reports (and the stride of what it downloads) has to
    grow when the reference's would, not when the device runs out of planes."""
    fem = _engine()
    nx, ny = 48, 40
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2, 4)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2, 4)
    trench = np.zeros((nx, ny), np.uint8); trench[20:28, :] = 1
    gm = g.mask_from_raster(trench.reshape(-1)); om = trench.reshape(-1)
    seen = []
    for k, (kind, m, mat, t) in enumerate([("d", None, 3, 10), ("d", None, 4, 5), ("d", None, 5, 5), ("e", 1, 255, 12), ("s", None, 4, 0),
                                           ("d", None, 6, 2), ("d", None, 7, 2), ("d", None, 8, 2), ("e", None, 8, 1), ("d", None, 6, 1),
                                           ("d", None, 7, 1)]):
        for grid, mk in ((g, gm if m else None), (o, om if m else None)):
            if kind == "d":
                grid.deposit_general(mk, mat, t, 0)
            elif kind == "e":
                grid.etch_general(mk, mat, t, 0)
            else:
                grid.strip_connected(mat)
        assert g.Lmax == o.Lmax, f"step {k}: Lmax {g.Lmax} vs {o.Lmax}"
        _assert_same(g, o.arrays(), f"step {k}")
        seen.append(o.Lmax)
    assert seen[0] == 4 and seen[-1] > 4 and (o.arrays()[0].max() < g.Lmax + 1)
    g.snapshot_save(3)
    g.deposit_general(None, 3, 1, 0); g.snapshot_load(3)
    assert g.Lmax == o.Lmax
    _assert_same(g, o.arrays(), "after snapshot")
    g.close()


def test_first_operation_after_a_reset():
    """createNewGrid does not clear the column streams; whatever comes first on the reset grid has to
    see empty columns: an unmasked deposit (which defines the streams itself), or anything else."""
    fem = _engine()
    nx, ny = 40, 36
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    half = np.zeros((nx, ny), np.uint8); half[:, :17] = 1
    gm, om = g.mask_from_raster(half.reshape(-1)), half.reshape(-1)

    def dirty():
        for grid, m in ((g, gm), (o, om)):
            grid.deposit_general(None, 3, 7, 0); grid.deposit_general(m, 4, 3, 0); grid.deposit_general(None, 5, 4, 1.0)
            grid.etch_general(m, 255, 6, 0)

    firsts = {
        "unmasked deposit": lambda grid, m: grid.deposit_general(None, 6, 2.5, 0),
        "masked deposit": lambda grid, m: grid.deposit_general(m, 6, 2.5, 0),
        "conformal deposit": lambda grid, m: grid.deposit_general(m, 6, 4, 1.0),
        "etch": lambda grid, m: grid.etch_general(None, 255, 3, 0),
        "lateral etch": lambda grid, m: grid.etch_general(None, 255, 4, 1.0),
        "cmp": lambda grid, m: grid.cmp(2.0, 0),
        "strip": lambda grid, m: grid.strip_connected(4),
        "cavity pass": lambda grid, m: grid.identify_cavity(),
        "wet etch": lambda grid, m: grid.etch_wet(255, 2.0),
        "ALD": lambda grid, m: grid.deposit_ALD(6, 2.0),
        "nothing": lambda grid, m: None,
    }
    for name, first in firsts.items():
        dirty()
        g.createNewGrid(); o.reset()
        first(g, gm); first(o, om)
        assert g.maxHeight() == o.max_height(), name
        assert g.checksum() == o.checksum(), name
        _assert_same(g, o.arrays(), name)
        g.deposit_general(gm, 7, 1.5, 0); o.deposit_general(om, 7, 1.5, 0)        # and life goes on
        g.etch_general(None, 7, 1.0, 0); o.etch_general(None, 7, 1.0, 0)
        assert g.checksum() == o.checksum(), name + " / next steps"
    # accessors, snapshots and uploads on a reset grid
    dirty(); g.createNewGrid(); o.reset()
    g.addLayer(3, 4, 5, 0.0, 2.0); o.addLayer(3, 4, 5, 0.0, 2.0)
    assert g.checksum() == o.checksum()
    dirty(); g.createNewGrid(); o.reset()
    g.snapshot_save(1); g.deposit_general(None, 3, 5, 0); g.snapshot_load(1)
    assert g.checksum() == o.checksum()
    dirty(); ln, mat, zp = o.arrays(); lm = o.Lmax
    g.createNewGrid(); g.upload(ln, mat, zp, lm)
    assert g.checksum() == o.checksum()
    g.close()


def test_material_2_laid_down_as_a_film():
    """Material id 2 is what identify_cavity strips: a flow that deposits it like any film must take
    the full cavity pass after conformal deposits, not the no-gap shortcut."""
    fem = _engine()
    nx, ny = 64, 48
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    half = np.zeros((nx, ny), np.uint8); half[:, :20] = 1
    pit = np.zeros((nx, ny), np.uint8); pit[24:40, 10:30] = 1
    gm = {"half": g.mask_from_raster(half.reshape(-1)), "pit": g.mask_from_raster(pit.reshape(-1))}
    om = {"half": half.reshape(-1), "pit": pit.reshape(-1)}
    for k, (kind, m, mat, t, S) in enumerate([("d", None, 3, 10, 0), ("d", "half", 2, 3, 0), ("d", None, 4, 4, 1.0), ("e", "pit", 255, 6, 0),
                                              ("d", None, 5, 2, 1.0), ("d", "pit", 2, 2, 0.9), ("e", None, 5, 3, 1.0), ("d", None, 6, 3, 1.0)]):
        for grid, masks in ((g, gm), (o, om)):
            fn = grid.deposit_general if kind == "d" else grid.etch_general
            fn(masks[m] if m else None, mat, t, S)
        assert g.checksum() == o.checksum(), f"step {k}"
    _assert_same(g, o.arrays(), "final")
    g.close()


@pytest.mark.parametrize("budget", [32, None])
def test_many_deposit_steps_recycle_planes(budget):
    """Every depositing launch takes a plane of its own.  With a budget of 32 planes and 90 deposits on
    short columns the engine has to compact the columns several times; without one it maps more planes
    behind the old ones.  Oracle equality after every step, snapshots across a compaction / growth included."""
    fem = _engine()
    nx, ny = 72, 64
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    if budget:
        g.set_plane_budget(budget)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    rng = _lcg(9)
    masks = []
    for _ in range(3):
        red = fem.reduce_mask(_random_mask_data(rng, 2 * nx, 9))
        masks.append((g.mask_from_shapes(red), o.mask_raster(red)))
    saved = None
    for k in range(90):
        gm, om = masks[k % 3] if k % 4 else (None, None)
        mat = 3 + k % 4
        S = [0, 0, 1.0, 0.5][k % 4]
        g.deposit_general(gm, mat, 1.5 + (k % 5), S); o.deposit_general(om, mat, 1.5 + (k % 5), S)
        assert g.checksum() == o.checksum(), f"deposit {k}"
        gm, om = masks[(k + 1) % 3]
        g.etch_general(gm, 255, 2.0 + (k % 7), [0, 1.0, 0][k % 3]); o.etch_general(om, 255, 2.0 + (k % 7), [0, 1.0, 0][k % 3])
        assert g.checksum() == o.checksum(), f"etch {k}"
        if k == 20:
            g.snapshot_save(7); saved = o.arrays() + (o.Lmax,)
        if k == 60:                                   # back to the state of step 20, then on
            g.snapshot_load(7); o.upload(*saved)
            assert g.checksum() == o.checksum()
    assert g.Lmax == o.Lmax == 32
    _assert_same(g, o.arrays(), "final")
    # the incremental models size their scratch planes by the planes in use: the grid gives back what
    # the 90 deposits mapped beyond that, then works out of place, then takes deposits again
    g.etch_wet(255, 3.0); o.etch_wet(255, 3.0)
    assert g.checksum() == o.checksum(), "etch_wet after the long run"
    g.deposit_ALD(6, 2.5); o.deposit_ALD(6, 2.5)
    assert g.checksum() == o.checksum(), "deposit_ALD after the long run"
    for k in range(40):
        g.deposit_general(masks[k % 3][0], 3 + k % 3, 1.0, 0); o.deposit_general(masks[k % 3][1], 3 + k % 3, 1.0, 0)
    assert g.Lmax == o.Lmax
    _assert_same(g, o.arrays(), "after trimming and growing again")
    g.close()


def test_columns_taller_than_64_intervals():
    """Stacks deeper than the split form's links reach (64 intervals): the engine keeps such a grid in whole form
    and runs vertical and lateral steps, growth of Lmax included, through the one-column kernels; when etches
    bring the stacks down again it returns to the split form.  Every step against the oracle."""
    fem = _engine()
    nx, ny = 40, 36
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    rng = _lcg(77)
    red = fem.reduce_mask(_random_mask_data(rng, 2 * nx, 6))
    gm, om = g.mask_from_shapes(red), o.mask_raster(red)
    def both(fn, *a):
        getattr(g, fn)(*[gm if x is om else x for x in a]); getattr(o, fn)(*a)
    for k in range(100):
        mask = om if k % 5 == 3 else None
        S = 1.0 if k % 7 == 6 else 0.0                     # now and then a conformal deposit (R = 1)
        both("deposit_general", mask, 3 + k % 2, 1.0 + 0.1 * (k % 4) + (1.2 if S else 0), S)
        assert g.checksum() == o.checksum(), f"deposit {k}"
        assert g.Lmax == o.Lmax, f"deposit {k}"
        if k % 9 == 8:
            both("etch_general", om, 255, 0.7, 0.0 if k % 2 else 1.0)   # partial cuts, vertical and lateral (t*S = 0.7 < dx: vertical either way)
            assert g.checksum() == o.checksum(), f"etch {k}"
        if k == 80:
            both("etch_general", None, 3 + k % 2, 2.5, 1.0)            # a lateral etch (R = 2) on the tall stacks
            assert g.checksum() == o.checksum(), f"lateral etch {k}"
    assert o.arrays()[0].max() > 70 and g.Lmax == o.Lmax >= 80
    _assert_same(g, o.arrays(), "tall stacks")
    # down again: whole layers go, the stacks fit the links again, and the fast kernels take over
    both("etch_general", None, 255, 60.0, 0.0)
    assert g.checksum() == o.checksum()
    for k in range(6):
        both("deposit_general", None if k % 2 else om, 5 + k % 2, 2.0, 1.0 if k == 3 else 0.0)
        assert g.checksum() == o.checksum(), f"after the etch, deposit {k}"
    _assert_same(g, o.arrays(), "after coming down")
    g.close()


def test_vertical_steps_at_awkward_thicknesses():
    """Vertical deposits and etches run on integer height steps when the host can vouch for every case of a thickness
    (statistics off, as here).  Thicknesses at and around the places where that could go wrong -- half a height step,
    a whole number of steps plus or minus the reference's 1e-9 and fractions of it, a layer etched by exactly its own
    thickness and a hair more or less -- against the oracle, step by step."""
    fem = _engine()
    nx, ny = 48, 64
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    rng = _lcg(4242)
    red = fem.reduce_mask(_random_mask_data(rng, 2 * nx, 5))
    gm, om = g.mask_from_shapes(red), o.mask_raster(red)
    def both(fn, mask, *a):
        getattr(g, fn)(gm if mask else None, *a); getattr(o, fn)(om if mask else None, *a)
    e = 1e-9
    thick = [10.0, 0.05, 0.15, 0.25, 7.25, 7.25 + e, 7.25 - e, 10.0 + e, 10.0 - e, 10.0 + e / 10, 10.0 - e / 10, 10.0 + 2 * e,
             3.3333333333, 0.1 + e, 0.1 - e, 0.1, 2 * e, e, e / 2, 1e-7, 123.456, 0.30000000000000004, 1.0 / 3.0]
    k = 0
    for t in thick:
        for mask in (False, True):
            both("deposit_general", mask, 3 + k % 4, t, 0.0); k += 1
            assert g.checksum() == o.checksum(), f"deposit {t!r} mask={mask}"
    for t in thick:
        # a fresh 10 nm layer of material 7 everywhere (20 nm where the mask is open), then the etch under test
        both("deposit_general", False, 7, 10.0, 0.0)
        both("deposit_general", True, 7, 10.0, 0.0)
        for mask, mat in ((True, 7), (False, 255), (False, 7)):
            both("etch_general", mask, mat, t, 0.0)
            assert g.checksum() == o.checksum(), f"etch {t!r} mask={mask} mat={mat}"
    _assert_same(g, o.arrays(), "final")
    g.close()


def test_strip_connected_along_a_serpentine():
    """A buried layer of material 4 that winds through the grid (full rows joined at alternating ends) and is open
    to the air at one end only: the flood has to run the whole path, against and along the sweep direction, over many
    batches of sweeps.  A second, closed loop of the same material must stay.  Against the oracle."""
    fem = _engine()
    nx, ny, lm = 41, 48, 32
    n = nx * ny
    ln = np.full(n, 3, np.uint8)
    mat = np.zeros((n, lm), np.uint8); zp = np.zeros((n, lm, 2), np.uint16)
    mat[:, 0] = 3; zp[:, 0] = (0, 100)
    mat[:, 1] = 6; zp[:, 1] = (100, 150)
    mat[:, 2] = 5; zp[:, 2] = (150, 200)
    M = mat.reshape(nx, ny, lm)
    for i in range(0, 33, 2):
        M[i, 2:ny - 2, 1] = 4                                  # the runs ...
        if i + 2 <= 32:
            M[i + 1, (ny - 3) if (i // 2) % 2 == 0 else 2, 1] = 4   # ... and the bends
    M[36:40, 10:20, 1] = 4; M[37:39, 11:19, 1] = 6             # a closed ring, never exposed
    ln.reshape(nx, ny)[0, 2] = 2                               # the cap is missing over the first cell of the path
    M[0, 2, 2] = 0; zp.reshape(nx, ny, lm, 2)[0, 2, 2] = (0, 0)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    g.upload(ln, mat.reshape(-1), zp.reshape(-1), lm); o.upload(ln, mat.reshape(-1), zp.reshape(-1), lm)
    g.strip_connected(4); o.strip_connected(4)
    oa = o.arrays()
    _assert_same(g, oa, "strip along the path")
    left = (oa[1].reshape(nx, ny, -1)[:, :, :3] == 4).any(axis=2)
    assert not left[:34].any() and left[36:40].any()           # the whole path went, the ring stayed
    g.close()


@pytest.mark.parametrize("seed,layers", [(11, 6), (12, 3), (13, 12)])
def test_strip_flood_by_components_equals_sweeps(seed, layers, monkeypatch):
    """strip_connected joins the pieces of the material in a union-find forest and hands the marks over the roots
    (a fixed number of launches); CG_STRIP_SWEEPS=1 relaxes sweep after sweep to the fixpoint instead.  Both against the
    oracle, on random many-layered columns (thin gaps, four materials: most pieces are small, some span the grid) and
    through identify_cavity, whose flood runs on material 2."""
    fem = _engine()
    nx, ny, lm = 40, 56, 32
    ln, mat, zp = _layered_columns(nx, ny, lm, layers, seed)
    grids = []
    for sweeps in (False, True):
        if sweeps:
            monkeypatch.setenv("CG_STRIP_SWEEPS", "1")
        else:
            monkeypatch.delenv("CG_STRIP_SWEEPS", raising=False)
        g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
        g.upload(ln, mat, zp, lm)
        g.strip_connected(4)
        g.identify_cavity()
        g.strip_connected(5)
        grids.append(g)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    o.upload(ln, mat, zp, lm)
    o.strip_connected(4); o.identify_cavity(); o.strip_connected(5)
    oa = o.arrays()
    _assert_same(grids[0], oa, "components")
    _assert_same(grids[1], oa, "sweeps")
    for g in grids:
        g.close()


def _layered_columns(nx, ny, lm, layers, seed):
    """Every column `layers` layers on a 5 nm pitch whose ends are pulled in by 0 .. 0.8 nm each (so most
    neighbouring layers are parted by a thin gap, at slightly different heights from column to column); one column
    in seven is a single solid layer of the same height instead."""
    rng = np.random.default_rng(seed)
    n, P = nx * ny, 50
    ln = np.full(n, layers, np.uint8)
    mat = np.zeros((n, lm), np.uint8); zp = np.zeros((n, lm, 2), np.uint16)
    for c in range(n):
        if rng.random() < 1 / 7:
            ln[c] = 1; mat[c, 0] = 3; zp[c, 0] = (0, layers * P)
            continue
        for k in range(layers):
            z0 = k * P + (int(rng.integers(0, 9)) if k > 0 else 0)
            z1 = (k + 1) * P - int(rng.integers(0, 9))
            mat[c, k] = int(rng.integers(3, 7)); zp[c, k] = (z0, z1)
    return ln, mat.reshape(-1), zp.reshape(-1)


@pytest.mark.parametrize("seed,layers", [(5, 15), (6, 2)])
def test_wet_etch_and_ald_on_many_layered_columns(seed, layers):
    """Uploaded columns of 15 layers with thin gaps between them (layers + gaps + the cavity pass's top piece stay within
    Lmax = 32, which the reference does not check, script_columngrid2.js:60-69): a column's 8 neighbours then throw far more removal
    intervals at it than the wet-etch kernel keeps per lane (it goes back to the neighbours' records for those
    columns); with 2 layers everything stays within the list.  The ALD coat meets tens of seeds per neighbour.
    Against the oracle on the full arrays."""
    fem = _engine()
    nx, ny, lm = 28, 24, 32
    ln, mat, zp = _layered_columns(nx, ny, lm, layers, seed)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    g.upload(ln, mat, zp, lm); o.upload(ln, mat, zp, lm)
    assert g.checksum() == o.checksum()
    g.etch_wet(255, 0.4, dt=0.2); o.etch_wet(255, 0.4, dt=0.2)
    _assert_same(g, o.arrays(), "etch_wet(ALL)")
    assert o.arrays()[0].max() >= layers                       # (the stacks are still there)
    g.etch_wet(4, 0.5); o.etch_wet(4, 0.5)
    _assert_same(g, o.arrays(), "etch_wet(4)")
    # the coat adds up to two pieces per gap: fewer layers, so that the columns stay within Lmax
    ln, mat, zp = _layered_columns(nx, ny, lm, min(layers, 6), seed + 100)
    g.upload(ln, mat, zp, lm); o.upload(ln, mat, zp, lm)
    g.deposit_ALD(6, 1.1, dt=0.6); o.deposit_ALD(6, 1.1, dt=0.6)
    _assert_same(g, o.arrays(), "deposit_ALD")
    g.etch_wet(6, 0.3); o.etch_wet(6, 0.3)
    _assert_same(g, o.arrays(), "etch_wet(6) after the coat")
    g.close()


@pytest.mark.parametrize("dx,zs,lmax", [(1.0, 0.1, 32), (2.0, 0.05, 32), (3.0, 0.25, 16), (0.5, 0.2, 32), (2.0, 0.1, 4), (2.0, 0.1, 1)])
def test_other_pitches_and_height_steps(dx, zs, lmax):
    """Cell pitch and height quantum other than the defaults (2 nm, 0.1 nm), small initial Lmax
    (1 -> 2: script_columngrid.js:302-304): every step kind against the oracle."""
    fem = _engine()
    nx, ny = 64, 48
    g = fem.ColumnGrid(dx * nx, dx * ny, dx, dx, lmax, zs)
    o = OracleGrid(dx * nx, dx * ny, dx, dx, lmax, zs)
    assert (g.NX, g.NY, g.Lmax) == (o.NX, o.NY, o.Lmax)
    rng = _lcg(int(dx * 10 + zs * 100))
    red = fem.reduce_mask(_random_mask_data(rng, int(dx * nx), 7))
    gm, om = g.mask_from_shapes(red), o.mask_raster(red)
    np.testing.assert_array_equal(gm.download(), om)
    t0 = 4 * dx
    steps = [("d", None, 3, 5 * t0, 0), ("d", None, 4, 1.3 * t0, 0), ("d", gm, 5, 2.1 * t0, 0), ("e", gm, 255, 3.3 * t0, 0),
             ("d", None, 6, 1.7 * t0, 1.0), ("e", None, 6, 0.9 * t0, 1.0), ("s", None, 4), ("d", None, 7, t0, 0.7),
             ("p", None, 0, 0.8 * t0), ("e", gm, 3, 1.2 * t0, 0.5), ("c",)]
    stopped = None
    for idx, st in enumerate(steps):
        kind = st[0]
        for grid, m in ((g, gm), (o, om)):
            mask = m if (len(st) > 1 and st[1] is not None) else None
            if kind == "d":
                grid.deposit_general(mask, st[2], st[3], st[4])
            elif kind == "e":
                grid.etch_general(mask, st[2], st[3], st[4])
            elif kind == "s":
                grid.strip_connected(st[2])
            elif kind == "p":
                grid.cmp(st[3], st[2])
            else:
                grid.identify_cavity()
        # Only a step that runs identify_cavity (lateral deposit, cmp, the pass itself) with a column that fills
        # every slot is undefined: the reference writes past the column there (no bound check, cg2:60-69).
        # Vertical deposits on full columns are well defined (the growth happens at the next append, cg1:418).
        ran_cavity = kind in ("p", "c") or (kind == "d" and st[3] * st[4] >= dx)
        if ran_cavity and o.arrays()[0].max() >= o.Lmax:
            stopped = idx
            break
        assert g.checksum() == o.checksum(), f"step {idx} {st}"
        assert g.Lmax == o.Lmax and g.maxHeight() == o.max_height(), f"step {idx}"
    if stopped is None:
        _assert_same(g, o.arrays(), "final")
    else:
        assert stopped >= 4, f"only {stopped} steps were compared before the reference left its arrays"
    g.close()


@pytest.mark.parametrize("nx,ny,seed,zs", [(96, 128, 21, 0.1), (61, 64, 22, 0.1), (128, 260, 23, 0.05), (40, 52, 24, 0.25), (72, 96, 25, 0.07)])
def test_runs_of_vertical_steps_in_one_launch(nx, ny, seed, zs):
    _check_runs_of_vertical_steps(nx, ny, seed, zs)


@pytest.mark.parametrize("nx,ny,seed", [(96, 160, 21), (128, 256, 22), (70, 324, 23), (64, 516, 24), (200, 68, 25), (33, 260, 26), (150, 512, 27), (90, 324, 28)])
def test_vertical_steps_inside_lateral_deposits(nx, ny, seed):
    _check_steps_inside_lateral_deposits(nx, ny, seed, 0)


@pytest.mark.parametrize("nshapes", [5, 40])
def test_polygon_with_points_outside_the_table_is_refused(nshapes):
    """A polygon record that names points outside the point table is refused before anything reads them: by the host's
    pass over a short list, by the binning kernel for a long one (from 8 shapes on).  The handle goes on working and
    the same list with the record repaired rasterises like the oracle's."""
    fem = _engine()
    nx, ny = 96, 128
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    rng = _lcg(77)
    data = _random_mask_data(rng, 2 * max(nx, ny), nshapes)
    red, plain = fem.reduce_mask(data, packed=True), fem.reduce_mask(data)      # (the oracle's wrapper reads the list form)
    sh = red["shapes"]
    poly = [k for k in range(len(sh)) if sh["type"][k] == fem.colgrid.SHAPE_POLYGON]
    assert poly, "the seeded list holds no polygon"
    k = poly[-1]
    good = int(sh["pt_off"][k])
    for bad in (len(red["pts"]) // 2 - int(sh["pt_cnt"][k]) + 1, -1):
        sh["pt_off"][k] = bad
        with pytest.raises(fem.ColGridError) as e:
            g.mask_from_shapes(red)
        assert "polygon points" in str(e.value) and f"shape {k}" in str(e.value)
    sh["pt_off"][k] = good
    m = g.mask_from_shapes(red)
    np.testing.assert_array_equal(m.download(), o.mask_raster(plain))
    g.deposit_general(m, 3, 5, 0); o.deposit_general(o.mask_raster(plain), 3, 5, 0)
    assert g.checksum() == o.checksum()
    g.close()


@pytest.mark.parametrize("nx,ny,seed,parts", [(512, 256, 31, 2), (768, 256, 32, 3), (640, 128, 33, 2), (600, 512, 34, 3), (520, 260, 35, 2)])
def test_vertical_steps_inside_lateral_deposits_on_slabs(nx, ny, seed, parts):
    """The same on a grid cut into slabs (cg_group_*): a slab's deposit waits as well, the steps behind it go out inside
    the launch over the rows that read no neighbour's message and in a pass of their own over the rows that do.  Sizes whose
    slabs have such inner rows (tiles are 64 rows tall) and whose edge bands are whole workgroups of columns; the last size
    has bands that are not, and takes the launches one by one."""
    _check_steps_inside_lateral_deposits(nx, ny, seed, parts)


def test_a_slabs_deposit_takes_the_steps_behind_it():
    """What the slab test above relies on, seen from the launch counters: on a slab with a neighbour on either side a
    lateral deposit followed by two vertical steps goes out as a launch of the deposit-with-steps kind (inner rows), plain
    deposit launches (the two bands that read messages) and two fused vertical launches (the steps on those bands)."""
    fem = _engine()
    n = 256
    g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(3 * n, n, n, n))
    def step(kind, mat, t, S):
        R = g.stencil_begin(kind, None, mat, t, S, 1, 1)
        if R:
            g.stencil_interior(); g.stencil_finish()
    step(0, 3, 30, 0)               # (0: deposit, 1: etch)
    g.sync()
    g.profile(True)
    step(0, 4, 7.5, 1.0)            # radius 4
    step(0, 5, 3.0, 0)
    step(1, 255, 1.0, 0)
    g.sync()
    prof = g.profile(False)
    assert prof.get("k_stencil<deposit+vertical>", (0, 0))[1] == 1, prof
    assert prof.get("k_stencil<deposit>", (0, 0))[1] == 1, prof             # (one timed scope around the two bands' launches)
    assert prof.get("k_vertical_fused", (0, 0))[1] == 2, prof
    # the run on a reset grid in front of a deposit without a mask: the rows the edge launches read get it when the step is
    # opened (one table launch per band), the inner rows inside the deposit's launch; no launch of the run on its own
    g.createNewGrid()
    g.sync()
    g.profile(True)
    step(0, 3, 30, 0)
    step(0, 4, 2.0, 0)
    step(0, 5, 9.5, 1.0)            # radius 5: takes the run in front along, nothing behind
    g.sync()
    prof = g.profile(False)
    assert prof.get("k_stencil<deposit+vertical>", (0, 0))[1] == 1, prof
    assert prof.get("k_stencil<deposit>", (0, 0))[1] == 1, prof
    assert prof.get("k_vertical_fused", (0, 0))[1] == 2, prof               # (the two table launches over the edge rows)
    assert "k_vertical_g4<deposit>" not in prof, prof
    g.close()


def _check_steps_inside_lateral_deposits(nx, ny, seed, parts):
    """A lateral deposit on an undivided grid waits for the vertical steps behind it and applies them inside its own launch
    (one, two or up to five of them: three instantiations), and a deposit without a mask that finds the queue of a reset grid
    waiting takes that run along too.  Seeded cases of both: radii 1..6 (5 and 6 take no steps behind them), lateral deposits
    with and without a mask, 0..7 vertical steps behind (more than five: the rest goes out as a run of its own), etches in
    the tail that remove what the deposit has just laid down and go on into older layers, strips whose last lanes lie beyond
    the grid (ny not a multiple of 256).  The state is compared with the oracle only after each case, so the steps of a case
    really go out together."""
    fem = _engine()
    rng = _lcg(seed)
    L = 2 * max(nx, ny)
    g = fem.GroupGrid(2 * nx, 2 * ny, 2, 2, devices=[0] * parts) if parts else fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    reds = [fem.reduce_mask(_random_mask_data(rng, L, 7)) for _ in range(3)]
    masks = [(g.mask_from_shapes(r), o.mask_raster(r)) for r in reds]
    ts = [0.05, 1.0, 2.5, 4, 6, 7.25, 8, 10, 13.05, 21]
    def both(fn, mi, *a):
        gm, om = (None, None) if mi is None else masks[mi]
        getattr(g, fn)(gm, *a); getattr(o, fn)(om, *a)
    def tail(n):
        for _ in range(n):
            kind = "deposit_general" if next(rng) % 5 < 2 else "etch_general"
            mi = next(rng) % 4
            mat = [3, 4, 5, 6, 255][next(rng) % (5 if kind == "etch_general" else 4)]
            both(kind, mi if mi < 3 else None, mat, ts[next(rng) % len(ts)], 0)
    for case in range(18):
        fresh = case % 3 == 0
        if fresh:                                              # a run on the reset grid in front of the deposit
            g.createNewGrid(); o.reset()
            both("deposit_general", None, 3, 30, 0)
            for k in range(next(rng) % 5):
                mi = next(rng) % 4
                both("deposit_general" if k % 2 == 0 else "etch_general", mi if mi < 3 else None, 4 + k % 2 if k % 2 == 0 else 255, ts[next(rng) % len(ts)], 0)
        R = 1 + next(rng) % 6
        S = [1.0, 0.5][next(rng) % 2]
        t = 2.0 * R / S - 0.5                                  # t*S just under R cells of 2 nm: radius R
        mi = None if (fresh or next(rng) % 3) else next(rng) % 3
        both("deposit_general", mi, 3 + next(rng) % 4, t, S)
        n = [0, 1, 1, 2, 2, 3, 5, 7][next(rng) % 8]
        if case % 5 == 4:                                      # an etch that takes the new layer off again and goes on below it
            both("etch_general", next(rng) % 3, 255, t + 3.0, 0); n = max(0, n - 1)
        tail(n)
        assert g.checksum() == o.checksum(), f"case {case}: R={R} S={S} mask={mi} fresh={fresh} tail={n}"
    _assert_same(g, o.arrays(), "after the cases")
    g.close()


def _check_runs_of_vertical_steps(nx, ny, seed, zs=0.1):
    """Consecutive vertical steps are queued and applied in one pass over the columns (k_vertical_fused) when something
    else looks at the grid.  Seeded runs of 1..9 vertical steps (the queue holds 6) with and without masks, thicknesses
    with and without an integer licence, extensions by the same material, etches that take several of the run's own
    layers and go on into older ones, blanket deposits on a reset grid (worked out on the host), a reset, a snapshot
    and the release of a mask while steps are queued -- the state is compared with the oracle only between runs, so the
    steps of a run really go out together.  One grid also holds gaps (STRIP) that the etches uncover."""
    fem = _engine()
    rng = _lcg(seed)
    L = 2 * max(nx, ny)
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2, 32, zs)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2, 32, zs)
    reds = [fem.reduce_mask(_random_mask_data(rng, L, 9)) for _ in range(3)]
    masks = [(g.mask_from_shapes(r), o.mask_raster(r)) for r in reds]
    ts = [0.04, 0.05, 1.0, 2.5, 4, 6, 7.25, 10, 13.05, 21, 26.05, 55]
    checked = 0
    g.profile(True)
    for run in range(14):
        n = 1 + next(rng) % 9
        if run in (0, 5, 9):                               # a reset with nothing (or a queue) pending, then blanket films
            g.createNewGrid(); o.reset()
            steps = [("d", None, 3 + k % 3, [30, 20, 2.5, 10, 7.25][k % 5]) for k in range(1 + next(rng) % 5)]
            if run == 9:
                steps[1:1] = [("d", None, steps[0][2], 4)]                             # the same material twice: an extension
        else:
            steps = []
        for _ in range(n):
            kind = "d" if next(rng) % 5 < 2 else "e"
            mi = next(rng) % 4
            mat = [3, 4, 5, 6, 255][next(rng) % (5 if kind == "e" else 4)]
            steps.append((kind, mi if mi < 3 else None, mat, ts[next(rng) % len(ts)]))
        for kind, mi, mat, t in steps:
            gm, om = (None, None) if mi is None else masks[mi]
            (g.deposit_general if kind == "d" else g.etch_general)(gm, mat, t, 0)
            (o.deposit_general if kind == "d" else o.etch_general)(om, mat, t, 0)
        if run == 3:                                       # a snapshot taken and the grid wrecked and restored, steps queued on both sides
            g.snapshot_save(5)
            g.etch_general(None, 255, 500, 0); g.deposit_general(None, 7, 3, 0)
            g.snapshot_load(5); g.snapshot_drop(5)
            g.deposit_general(masks[0][0], 7, 1.5, 0); o.deposit_general(masks[0][1], 7, 1.5, 0)
        if run == 6:                                       # a mask released and made again while steps that use it are queued
            g.etch_general(masks[2][0], 255, 3.3, 0); o.etch_general(masks[2][1], 255, 3.3, 0)
            masks[2][0].destroy()
            masks[2] = (g.mask_from_shapes(reds[2]), masks[2][1])
        if run == 7 and seed % 2 == 1:                     # gaps under the tops: a lateral deposit over strips leaves holes to uncover
            g.strip_connected(4); o.strip_connected(4)
            g.deposit_general(None, 6, 5, 1.0); o.deposit_general(None, 6, 5, 1.0)
            if o.arrays()[0].max() >= o.Lmax:
                break
        assert g.checksum() == o.checksum(), f"run {run}: {steps}"
        assert g.Lmax == o.Lmax and g.maxHeight() == o.max_height(), f"run {run}"
        checked += 1
    assert checked >= 8
    assert g.profile(False).get("k_vertical_fused", (0, 0))[1] >= 8              # (the runs did go out as fused launches)
    _assert_same(g, o.arrays(), "final")
    g.close()


@pytest.mark.parametrize("t_etch,S", [(12.0, 1.0), (10.0, 1.0), (16.4, 0.5)])
def test_isotropic_overetch_rows_without_neighbours(t_etch, S):
    """An isotropic etch deeper than the film it takes off: the streaming lateral kernel updates the rows at the head of
    a tile from their own columns alone (the result is the floor whatever the neighbours are, script_columngrid.js:725)
    and switches to the stencil at the first row where a column could end above its floor -- here the band of rows with a
    thicker film, which starts in the middle of a tile.  With t equal to the film's thickness the float32 candidates
    land on either side of the floor and the stencil decides, as in the reference.  Against the oracle, full arrays."""
    fem = _engine()
    nx, ny = 200, 520
    g = fem.ColumnGrid(2 * nx, 2 * ny, 2, 2)
    o = OracleGrid(2 * nx, 2 * ny, 2, 2)
    band = np.zeros((nx, ny), np.uint8); band[90:121, 40:400] = 1
    holes = np.ones((nx, ny), np.uint8); holes[10:30, 100:130] = 0; holes[150:160, :] = 0
    gm_band, gm_holes = g.mask_from_raster(band.reshape(-1)), g.mask_from_raster(holes.reshape(-1))
    for grid, mb, mh in ((g, gm_band, gm_holes), (o, band.reshape(-1), holes.reshape(-1))):
        grid.deposit_general(None, 3, 30.3, 0)
        grid.deposit_general(None, 5, 10, 0)
        grid.deposit_general(mb, 5, 8, 0)                  # the film is 18 nm thick in the band
        grid.etch_general(mh, 5, t_etch, S)                # masked: rows 150..159 closed, a closed patch
    _assert_same(g, o.arrays(), "masked over-etch")
    for grid in (g, o):
        grid.deposit_general(None, 4, 7.7, 0)
        grid.etch_general(None, 255, t_etch, S)            # ALL: the floor is 0, nothing ends there
        grid.etch_general(None, 3, 50, 0.3)                # takes the whole substrate off wherever it is on top
    _assert_same(g, o.arrays(), "ALL and substrate")
    g.close()


def test_unequal_pitch():
    """dx != dy.  Vertical steps match the oracle for any pitch.  Lateral steps: the reference pads its height map by
    Rx cells in both directions (script_columngrid.js:471, 666), so they are well defined while the taps reach no
    further along y than along x (Ry <= Rx: any dy > dx; dy < dx only at thicknesses where the radii coincide) and
    match the oracle there (the fixtures unequal_pitch / unequal_pitch_t pin the same against the reference); a step
    with Ry > Rx reads outside the pad in the reference and is refused, leaving the grid as it was."""
    fem = _engine()
    g = fem.ColumnGrid(120, 90, 2.0, 3.0)
    o = OracleGrid(120, 90, 2.0, 3.0)
    assert (g.NX, g.NY) == (o.NX, o.NY) == (60, 30)
    rng = _lcg(77)
    red = fem.reduce_mask(_random_mask_data(rng, 90, 6))
    gm, om = g.mask_from_shapes(red), o.mask_raster(red)
    np.testing.assert_array_equal(gm.download(), om)
    for mat, t in ((3, 10), (4, 6.5), (3, 2)):
        g.deposit_general(None, mat, t, 0); o.deposit_general(None, mat, t, 0)
    g.etch_general(gm, 255, 4.4, 0); o.etch_general(om, 255, 4.4, 0)
    g.deposit_general(None, 5, 1.9, 1.0); o.deposit_general(None, 5, 1.9, 1.0)     # t*S < min(dx, dy): still vertical
    assert g.checksum() == o.checksum()
    for k, (kind, m, mat, t, S) in enumerate([("d", None, 5, 8, 1.0), ("e", None, 5, 8, 1.0), ("d", True, 6, 13.3, 0.6), ("e", True, 255, 9.1, 0.9),
                                              ("d", None, 7, 41, 1.0), ("e", None, 7, 23, 0.5)]):
        for grid, mask in ((g, gm), (o, om)):
            (grid.deposit_general if kind == "d" else grid.etch_general)(mask if m else None, mat, t, S)
        assert g.checksum() == o.checksum(), f"lateral step {k}"
    _assert_same(g, o.arrays(), "dx 2, dy 3")
    g.close()
    # dy < dx: fine while ceil(t/dy) == ceil(t/dx), refused beyond
    g = fem.ColumnGrid(120, 90, 3.0, 2.0)
    o = OracleGrid(120, 90, 3.0, 2.0)
    for mat, t in ((3, 10), (4, 6.5)):
        g.deposit_general(None, mat, t, 0); o.deposit_general(None, mat, t, 0)
    g.deposit_general(None, 5, 4, 1.0); o.deposit_general(None, 5, 4, 1.0)         # Rx = Ry = 2
    g.etch_general(None, 5, 2, 1.0); o.etch_general(None, 5, 2, 1.0)               # Rx = Ry = 1
    assert g.checksum() == o.checksum()
    with pytest.raises(fem.ColGridError):
        g.deposit_general(None, 5, 8, 1.0)                                         # Rx = 3, Ry = 4
    with pytest.raises(fem.ColGridError):
        g.etch_general(None, 5, 8, 1.0)
    assert g.checksum() == o.checksum()
    g.deposit_general(None, 6, 3.5, 0.8); o.deposit_general(None, 6, 3.5, 0.8)     # the handle still works
    _assert_same(g, o.arrays(), "dx 3, dy 2")
    g.close()


def test_upload_download_roundtrip_and_slices():
    fem = _engine()
    case = load_case("stencil_small")
    ln, mat, zp, lm = case["dump_arrays"][-1]
    g = _grid_for(case)
    g.upload(ln, mat, zp, lm)
    _assert_same(g, (ln, mat, zp), "roundtrip")
    nx, ny = case["nx"], case["ny"]
    M = mat.reshape(nx, ny, lm); Z = zp.reshape(nx, ny, lm, 2); Ln = ln.reshape(nx, ny)
    sl, sm, sz = g.slices("YZ", [0, 5, nx - 1])
    for s, i in enumerate([0, 5, nx - 1]):
        np.testing.assert_array_equal(sl[s], Ln[i]); np.testing.assert_array_equal(sm[s], M[i]); np.testing.assert_array_equal(sz[s], Z[i])
    sl, sm, sz = g.slices("XZ", [3, ny - 1])
    for s, j in enumerate([3, ny - 1]):
        np.testing.assert_array_equal(sl[s], Ln[:, j]); np.testing.assert_array_equal(sm[s], M[:, j]); np.testing.assert_array_equal(sz[s], Z[:, j])
    # snapshots
    g.snapshot_save(7)
    g.etch_general(None, 255, 10, 0)
    assert g.checksum() != 0
    g.snapshot_load(7)
    _assert_same(g, (ln, mat, zp), "snapshot")
    o = OracleGrid(case["domain"]["LX"], case["domain"]["LY"], 2, 2)
    o.upload(ln, mat, zp, lm)
    assert g.checksum() == o.checksum()
    for pos in (-1e9, -64.0, -0.99, 0.0, 1.0, 2.999, 63.0, 1e9):
        for ax in (0, 1):
            assert g.index_from_pos(ax, pos) == o.index_from_pos(ax, pos)
    g.close()


@pytest.mark.parametrize("parts", [2, 3])
def test_slabs_on_one_gpu_equal_whole_grid(parts):
    """The HIP slab path (stencil_begin / halo buffers / stencil_finish) with P slabs on one device,
    halos moved by plain device copies: bit-identical to the undivided grid after every step."""
    import torch
    fem = _engine()
    from fem_amd import flows
    nxl, ny = 48, 136
    nxg = nxl * parts
    flow = flows.flow_c4()
    specs = flows.mask_specs(2.0 * nxg)
    whole = fem.ColumnGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    o = OracleGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    slabs = [fem.ColumnGrid(dx=2.0, dy=2.0, slab=(nxg, ny, p * nxl, nxl)) for p in range(parts)]
    cap = 16 * ny * 3
    bufs = [[torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(4)] for _ in range(parts)]
    for s, b in zip(slabs, bufs):
        s.set_halo_buffers(*[t.data_ptr() for t in b], cap)
    wm, om, sm = {}, {}, [dict() for _ in range(parts)]
    for name in sorted({s["mask"] for s in flow if s["mask"]}):
        red = fem.reduce_mask(flows.shapes_of(specs[name], 2 * nxg, 2 * ny))
        wm[name] = whole.mask_from_shapes(red)
        om[name] = o.mask_raster(red)
        for p in range(parts):
            sm[p][name] = slabs[p].mask_from_shapes(red)          # rasterised with the slab's global row offset
            np.testing.assert_array_equal(sm[p][name].download(), om[name].reshape(nxg, ny)[p * nxl:(p + 1) * nxl].reshape(-1))
    for k, st in enumerate(flow):
        flows.apply_step(whole, st, wm)
        flows.apply_step(o, st, om)
        kind = 1 if st["kind"] == "ETCH" else 0
        S = 0.0 if st["kind"] == "SUBSTR" else st["S"]
        R = [slabs[p].stencil_begin(kind, None if st["mask"] is None else sm[p][st["mask"]], st["mat"], st["t"], S,
                                    p > 0, p < parts - 1) for p in range(parts)]
        if R[0]:
            n = R[0] * ny * 3
            if k % 2:                                   # rows that need no halo first, before any halo has arrived
                for s in slabs:
                    s.stencil_interior()
            for s in slabs:
                s.sync()
            for p in range(parts):
                if p > 0:
                    bufs[p][2][:n].copy_(bufs[p - 1][1][:n])       # my lower halo <- lower neighbour's upper rows
                if p < parts - 1:
                    bufs[p][3][:n].copy_(bufs[p + 1][0][:n])
            torch.cuda.synchronize()
            for s in slabs:
                s.stencil_finish()
        total = sum(s.checksum() for s in slabs) & 0xFFFFFFFFFFFFFFFF
        assert total == whole.checksum() == o.checksum(), f"step {k} {st}"
    for s in slabs + [whole]:
        s.close()


class _ThreadComm:
    """fem_amd.slabs communicator for P ranks that are threads of one process (one GPU)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def view(self, rank):
        comm = self

        class _Rank:
            def start_swap(self, send_lower, send_upper, recv_lower, recv_upper):
                import torch
                torch.cuda.synchronize()
                comm.slots[rank] = (send_lower, send_upper)
                comm.barrier.wait()
                return (recv_lower, recv_upper)

            def finish_swap(self, reqs):
                import torch
                recv_lower, recv_upper = reqs
                if rank > 0:
                    recv_lower.copy_(comm.slots[rank - 1][1])
                if rank < comm.world - 1:
                    recv_upper.copy_(comm.slots[rank + 1][0])
                torch.cuda.synchronize()
                comm.barrier.wait()

            def swap(self, *bufs):
                self.finish_swap(self.start_swap(*bufs))

            def reduce(self, values, op="max", dtype=None):
                comm.slots[rank] = list(values)
                comm.barrier.wait()
                cols = list(zip(*comm.slots))
                out = [max(c) if op == "max" else sum(c) for c in cols]
                comm.barrier.wait()
                return out

        return _Rank()


@pytest.mark.parametrize("parts,flow_name", [(2, "cs"), (3, "cs"), (3, "c4"), (2, "cwet"), (3, "cwet"), (2, "cw"), (3, "cw")])
def test_slab_driver_on_one_gpu(parts, flow_name):
    """fem_amd.slabs.SlabGrid itself (halo rows, all-reduced scalars, ghost-row flood of STRIP /
    identify_cavity, CMP) with P slabs driven by P threads on one device: the summed checksum equals
    the undivided grid's and the oracle's after every step."""
    from fem_amd import flows
    nxl = 40
    _run_slab_flow(parts, flows.FLOWS[flow_name](), flows.mask_specs(2.0 * nxl * parts), nxl, 72)


def _run_slab_flow(parts, flow, specs, nxl, ny):
    import threading
    import torch
    fem = _engine()
    from fem_amd import flows
    from fem_amd.slabs import SlabGrid
    nxg = nxl * parts
    o = OracleGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    names = sorted({s["mask"] for s in flow if s["mask"]})
    reds = {n: fem.reduce_mask(flows.shapes_of(specs[n], 2 * nxg, 2 * ny)) for n in names}
    om = {n: o.mask_raster(reds[n]) for n in names}
    want = []
    for st in flow:
        flows.apply_step(o, st, om)
        want.append((o.checksum(), o.max_height()))
    comm = _ThreadComm(parts)
    errors = []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(nxg, ny, rank * nxl, nxl))
            masks = {n: g.mask_from_shapes(reds[n]) for n in names}
            slab = SlabGrid(g, rank, parts, device=torch.device("cuda", 0), ny=ny, halo_rows=4, comm=comm.view(rank))
            for k, st in enumerate(flow):
                flows.apply_step(slab, st, masks)
                got = (slab.checksum(), slab.maxHeight())
                if got != want[k]:
                    raise AssertionError(f"rank {rank} step {k} {st}: {got} != {want[k]}")
            g.close()
        except BaseException as e:        # noqa: BLE001 - reported by the main thread
            errors.append(e)
            comm.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors[0]


def test_slab_ald_when_one_slab_holds_far_more_seeds():
    """deposit_ALD over two slabs of which only the upper one is patterned: its columns hold many seeds, the
    flat slab's one each, so the flat slab has to widen its seed arrays before the boundary rows' seeds arrive;
    the coat (3 rows of reach) crosses the slab edge in both directions."""
    from fem_amd import flows
    nxl, ny = 24, 40
    L = 2.0 * nxl * 2
    specs = {"Mup": {"kind": "shapes", "data": {"objects": [
        {"type": "rect", "left": 0, "top": 0, "width": 4 * L, "height": 4 * L, "angle": 0, "scaleX": 1, "scaleY": 1,
         "visible": True, "data": {"polarity": "positive"}},
        {"type": "rect", "left": L / 2 + 0.5, "top": 10, "width": 14, "height": 30, "angle": 0, "scaleX": 1, "scaleY": 1,
         "visible": True, "data": {"polarity": "negative"}},
        {"type": "circle", "left": L / 2 + 20, "top": 44, "radius": 9, "width": 18, "height": 18, "angle": 0, "scaleX": 1, "scaleY": 1,
         "visible": True, "data": {"polarity": "negative"}}]}}}
    s = flows._s
    flow = [s("SUBSTR", None, 3, 20), s("DEPO", None, 4, 6), s("DEPO", None, 5, 5), s("DEPO", None, 4, 4), s("DEPO", None, 6, 5),
            s("ETCH", "Mup", 255, 17), s("WETETCH", None, 4, 3), s("ALD", None, 7, 5.5), s("ALD", None, 5, 2)]
    _run_slab_flow(2, flow, specs, nxl, ny)


def test_slab_ald_refuses_a_slab_thinner_than_the_coat_reach():
    fem = _engine()
    g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(8, 16, 2, 2))
    g.deposit_general(None, 3, 10, 0)
    import torch
    ghost = torch.zeros(g.ghost_bytes(4), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception, match="reaches"):
        g.ald_collect_seeds_slab(7.0, 100, 4, ghost.data_ptr(), ghost.data_ptr())
    g.close()


def test_slab_driver_when_a_mask_id_is_handed_out_again():
    """A mask that opens no cell anywhere makes a lateral deposit return early -- after the reference has put the
    step's kernel into its kernel cache (script_columngrid.js:470 vs 485).  The mask is then destroyed and the next
    raster, an open one, gets its id: the slab driver must ask again whether it is open, and the next lateral
    deposit, whose cache key collides with the skipped step's (thicknesses closer than 1e-3), must find that
    step's kernel, as on the undivided grid."""
    import threading
    import torch
    fem = _engine()
    from fem_amd.slabs import SlabGrid
    parts, nxl, ny = 2, 24, 40
    nxg = nxl * parts
    o = OracleGrid(2.0 * nxg, 2.0 * ny, 2.0, 2.0)
    closed = np.zeros(nxg * ny, np.uint8)
    rng = np.random.RandomState(11)
    opened = (rng.rand(nxg * ny) < 0.7).astype(np.uint8)
    want = []
    o.deposit_general(None, 3, 20, 0); want.append(o.checksum())
    o.deposit_general(closed, 4, 6.0004, 1.0); want.append(o.checksum())
    o.deposit_general(opened, 5, 6.0001, 1.0); want.append(o.checksum())
    o.etch_general(opened, 5, 3, 1.0); want.append(o.checksum())
    assert want[0] == want[1] != want[2]
    comm = _ThreadComm(parts)
    errors = []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(nxg, ny, rank * nxl, nxl))
            mine = lambda r: np.ascontiguousarray(r.reshape(nxg, ny)[rank * nxl:(rank + 1) * nxl]).reshape(-1)
            slab = SlabGrid(g, rank, parts, device=torch.device("cuda", 0), ny=ny, halo_rows=4, comm=comm.view(rank))
            got = []
            slab.deposit_general(None, 3, 20, 0); got.append(slab.checksum())
            m0 = g.mask_from_raster(mine(closed))
            slab.deposit_general(m0, 4, 6.0004, 1.0); got.append(slab.checksum())
            old_id = m0.id
            m0.destroy()
            m1 = g.mask_from_raster(mine(opened))
            assert m1.id == old_id
            slab.deposit_general(m1, 5, 6.0001, 1.0); got.append(slab.checksum())
            slab.etch_general(m1, 5, 3, 1.0); got.append(slab.checksum())
            if got != want:
                raise AssertionError(f"rank {rank}: {got} != {want}")
            g.close()
        except BaseException as e:        # noqa: BLE001 - reported by the main thread
            errors.append(e)
            comm.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors[0]


@pytest.mark.parametrize("parts,flow_name", [(2, "c4"), (3, "c4"), (2, "cs"), (3, "cs"), (2, "cwet"), (3, "cw"), (1, "cs")])
def test_group_driven_by_the_library(parts, flow_name):
    """cg_group_*: the slab decomposition driven below the C ABI (one process, `parts` slabs on one device, rows moved
    by device copies on per-slab transfer streams): the same checksum, maximum height and Lmax as the oracle after
    every step, and the whole grid's arrays at the end."""
    fem = _engine()
    from fem_amd import flows
    nx, ny = 40 * parts + (1 if parts == 3 else 0), 72           # 121 rows over 3 slabs: uneven slabs
    flow = flows.FLOWS[flow_name]()
    specs = flows.mask_specs(2.0 * nx)
    g = fem.GroupGrid(2.0 * nx, 2.0 * ny, 2.0, 2.0, devices=[0] * parts)
    o = OracleGrid(2.0 * nx, 2.0 * ny, 2.0, 2.0)
    gm, om = {}, {}
    for name in sorted({s["mask"] for s in flow if s["mask"]}):
        red = fem.reduce_mask(flows.shapes_of(specs[name], 2 * nx, 2 * ny))
        gm[name] = g.mask_from_shapes(red)
        om[name] = o.mask_raster(red)
        np.testing.assert_array_equal(gm[name].download(), om[name])
    for k, st in enumerate(flow):
        flows.apply_step(g, st, gm)
        flows.apply_step(o, st, om)
        o.slider_reset()
        assert g.checksum() == o.checksum(), f"step {k} {st}"
        assert g.maxHeight() == o.max_height() and g.Lmax == o.Lmax, f"step {k} {st}"
    ln, mat, zp = g.download()
    oln, omat, ozp = o.arrays()
    np.testing.assert_array_equal(ln, oln); np.testing.assert_array_equal(mat, omat); np.testing.assert_array_equal(zp, ozp)
    # cross-sections of the whole domain, both ways, against the downloaded arrays
    L = g.Lmax
    idx = [0, nx // 2, nx - 1]
    sl, sm, sz = g.slices("YZ", idx)
    for q, i in enumerate(idx):
        np.testing.assert_array_equal(sl[q], ln.reshape(nx, ny)[i]); np.testing.assert_array_equal(sm[q], mat.reshape(nx, ny, L)[i])
    jdx = [1, ny - 2]
    sl, sm, sz = g.slices("XZ", jdx)
    for q, j in enumerate(jdx):
        np.testing.assert_array_equal(sl[q], ln.reshape(nx, ny)[:, j]); np.testing.assert_array_equal(sz[q], zp.reshape(nx, ny, L, 2)[:, j])
    if parts > 1:
        rccl, exchanges, moved = g.transport()
        assert not rccl and exchanges > 0 and moved > 0            # slabs sharing a device: device copies
    g.close()


def test_group_when_a_mask_id_is_handed_out_again():
    """The group's own 'does this mask open a cell anywhere' answer dies with the mask: the id of a destroyed all-closed
    mask goes to an open one, and the lateral deposit through it must happen -- with the kernel the skipped step left in
    the kernel cache (thicknesses closer than 1e-3 share a cache key, script_columngrid.js:326, 470)."""
    fem = _engine()
    nx, ny = 48, 40
    g = fem.GroupGrid(2.0 * nx, 2.0 * ny, 2.0, 2.0, devices=[0, 0])
    o = OracleGrid(2.0 * nx, 2.0 * ny, 2.0, 2.0)
    closed = np.zeros(nx * ny, np.uint8)
    opened = (np.random.RandomState(11).rand(nx * ny) < 0.7).astype(np.uint8)
    g.deposit_general(None, 3, 20, 0); o.deposit_general(None, 3, 20, 0)
    m0 = g.mask_from_raster(closed)
    g.deposit_general(m0, 4, 6.0004, 1.0); o.deposit_general(closed, 4, 6.0004, 1.0)
    assert g.checksum() == o.checksum()
    old = m0.id
    m0.destroy()
    m1 = g.mask_from_raster(opened)
    assert m1.id == old
    g.deposit_general(m1, 5, 6.0001, 1.0); o.deposit_general(opened, 5, 6.0001, 1.0)
    assert g.checksum() == o.checksum()
    g.etch_general(m1, 5, 3, 1.0); o.etch_general(opened, 5, 3, 1.0)
    assert g.checksum() == o.checksum()
    g.close()
