"""Synthetic process flows and masks of the benchmark configurations.

Everything is a deterministic function of the domain size; no RNG.  All configurations use
dx = dy = 2 nm, zScale 0.1, Lmax 32 and the materials Si 3 / Ox 4 / Nit 5 / 6 / 7.  A step is a
dict {kind: SUBSTR|DEPO|ETCH, mask: name|None, mat, t, S} dispatched the way the reference
runtime dispatches its process records (template/script_runtime.js:162-180): SUBSTR and DEPO
-> deposit_general (SUBSTR with conformality 0), ETCH -> etch_general.

Masks are given in nanometres so they scale with the domain.  Periodic masks have up to
hundreds of thousands of identical shapes on the large grids; for those the raster is also
available in closed form (integer arithmetic on the cell centres x = 2i+1, y = 2j+1).  `shapes_of`
expands the same mask into the fabric-style object list the reference's predicate consumes: the
benchmark rasterises that list and checks the result against the closed form.
"""
import numpy as np

SI, OX, NIT, M6, M7, ALL = 3, 4, 5, 6, 7, 255
DX = DY = 2.0


# ---------------------------------------------------------------------------------------------
# masks
# ---------------------------------------------------------------------------------------------
def mask_specs(L_nm):
    """Mask name -> spec for a square domain of L_nm nanometres (x extent; y may differ)."""
    return {
        # M: periodic positive (blocking) squares, pitch 64 nm, 32 x 32 nm
        "M": {"kind": "periodic_rects", "pitch": 64, "size": 32, "shift": 0},
        # M': M shifted by 16 nm
        "Mshift": {"kind": "periodic_rects", "pitch": 64, "size": 32, "shift": 16},
        # Mc: periodic blocking discs, radius 12 nm, pitch 48 nm
        "Mc": {"kind": "periodic_circles", "pitch": 48, "radius": 12},
        # Mr: the field is blocked except a large window rotated by 30 degrees
        "Mr": {"kind": "shapes", "data": {"objects": [
            {"type": "rect", "left": 0, "top": 0, "width": 4 * L_nm, "height": 4 * L_nm, "angle": 0, "scaleX": 1, "scaleY": 1,
             "visible": True, "data": {"polarity": "positive"}},
            {"type": "rect", "left": 0.30 * L_nm, "top": 0.05 * L_nm, "width": 0.55 * L_nm, "height": 0.35 * L_nm, "angle": 30,
             "scaleX": 1, "scaleY": 1, "visible": True, "data": {"polarity": "negative"}},
        ]}},
        # C1's three-shape mask scaled to the domain (256 columns <-> 512 nm)
        "M1": {"kind": "shapes", "data": {"objects": [
            {"type": "rect", "left": 100 * L_nm / 512, "top": 100 * L_nm / 512, "width": 200 * L_nm / 512, "height": 120 * L_nm / 512,
             "angle": 0, "scaleX": 1, "scaleY": 1, "visible": True, "data": {"polarity": "positive"}},
            {"type": "circle", "left": 300 * L_nm / 512, "top": 300 * L_nm / 512, "radius": 60 * L_nm / 512, "width": 120 * L_nm / 512,
             "height": 120 * L_nm / 512, "angle": 0, "scaleX": 1, "scaleY": 1, "visible": True, "data": {"polarity": "positive"}},
            {"type": "rect", "left": 140 * L_nm / 512, "top": 130 * L_nm / 512, "width": 40 * L_nm / 512, "height": 40 * L_nm / 512,
             "angle": 30, "scaleX": 1, "scaleY": 1, "visible": True, "data": {"polarity": "negative"}},
        ]}},
    }


def shapes_of(spec, LX_nm, LY_nm, x_range=None):
    """Expand a mask spec into the fabric-style JSON ({'objects': [...]}) the reference consumes.
    x_range = (x0, x1) in nm keeps only the shapes of a periodic mask that reach that band (what one slab
    of a decomposed domain needs; they all have the same polarity, so leaving the others out changes nothing)."""
    if spec["kind"] == "shapes":
        return spec["data"]
    objs = []
    p = spec["pitch"]
    if spec["kind"] == "periodic_rects":
        s, sh = spec["size"], spec["shift"]
        for x in range(sh, int(LX_nm), p):
            if x_range is not None and (x + s < x_range[0] or x > x_range[1]):
                continue
            for y in range(sh, int(LY_nm), p):
                objs.append({"type": "rect", "left": x, "top": y, "width": s, "height": s, "angle": 0, "scaleX": 1,
                             "scaleY": 1, "visible": True, "data": {"polarity": "positive"}})
    elif spec["kind"] == "periodic_circles":
        r = spec["radius"]
        for x in range(0, int(LX_nm), p):
            if x_range is not None and (x + 2 * r < x_range[0] or x > x_range[1]):
                continue
            for y in range(0, int(LY_nm), p):
                objs.append({"type": "circle", "left": x, "top": y, "radius": r, "width": 2 * r, "height": 2 * r, "angle": 0,
                             "scaleX": 1, "scaleY": 1, "visible": True, "data": {"polarity": "positive"}})
    else:
        raise ValueError(spec["kind"])
    return {"objects": objs}


def _periodic_open(spec, X, Y, xp):
    """Closed form of a periodic mask on integer cell centres X (column vector), Y (row vector).
    xp = numpy or torch; returns a bool array, True = open (step applies)."""
    p = spec["pitch"]
    if spec["kind"] == "periodic_rects":
        s, sh = spec["size"], spec["shift"]
        bx = ((X - sh) % p <= s) & (X >= sh)
        by = ((Y - sh) % p <= s) & (Y >= sh)
        return ~(bx & by)
    if spec["kind"] == "periodic_circles":
        r = spec["radius"]
        u = X % p - r
        v = Y % p - r
        return ~((u * u + v * v) <= r * r)
    raise ValueError(spec["kind"])


def raster_numpy(spec, nx, ny, i0=0):
    """uint8 raster (1 = open) of a periodic mask for rows [i0, i0+nx) on the CPU."""
    X = (2 * (np.arange(nx, dtype=np.int64) + i0) + 1)[:, None]
    Y = (2 * np.arange(ny, dtype=np.int64) + 1)[None, :]
    return _periodic_open(spec, X, Y, np).astype(np.uint8).reshape(-1)


def raster_torch(spec, nx, ny, i0, device):
    """Same raster produced on the GPU (torch tensor, uint8, nx*ny)."""
    import torch
    X = (2 * (torch.arange(nx, dtype=torch.int64, device=device) + i0) + 1)[:, None]
    Y = (2 * torch.arange(ny, dtype=torch.int64, device=device) + 1)[None, :]
    return _periodic_open(spec, X, Y, torch).to(torch.uint8).reshape(-1).contiguous()


# ---------------------------------------------------------------------------------------------
# flows
# ---------------------------------------------------------------------------------------------
def _s(kind, mask, mat, t, S=0.0):
    return {"kind": kind, "mask": mask, "mat": mat, "t": float(t), "S": float(S)}


def flow_c1():
    """256^2: deposit -> pattern -> etch, one mask (the reference's own CPU-runnable case)."""
    return [_s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 20), _s("ETCH", "M1", OX, 20)]


def flow_c2():
    """2048^2: 10-step multi-material flow, anisotropic only (pure per-column)."""
    return [_s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 20), _s("DEPO", None, NIT, 10),
            _s("ETCH", "M", NIT, 10), _s("ETCH", "M", OX, 12), _s("DEPO", None, M6, 8),
            _s("ETCH", None, M6, 8), _s("DEPO", None, M7, 15), _s("ETCH", "Mshift", ALL, 10),
            _s("ETCH", "Mshift", SI, 5)]


def flow_c3():
    """8192^2: patterned stack, then conformal deposit (R=5, 81 taps) and isotropic etch (R=3, 29 taps)."""
    return [_s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 20), _s("ETCH", "M", OX, 20),
            _s("DEPO", None, NIT, 10, 1.0), _s("ETCH", None, NIT, 6, 1.0)]


def flow_c4():
    """16384^2: three masks, 20 steps alternating the C2 and C3 step kinds, t in 4..30, S in {0, .5, 1}."""
    return [
        _s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 20), _s("DEPO", None, NIT, 10),
        _s("ETCH", "M", NIT, 10), _s("ETCH", "M", OX, 12),
        _s("DEPO", None, NIT, 10, 1.0),            # conformal, R = 5
        _s("ETCH", None, NIT, 6, 1.0),             # isotropic, R = 3
        _s("DEPO", None, M6, 8, 0.5),              # R = 4, lateral reach 4 nm
        _s("ETCH", "Mc", M6, 8), _s("DEPO", None, M7, 15),
        _s("ETCH", "Mshift", ALL, 10), _s("ETCH", "Mshift", SI, 5),
        _s("DEPO", None, OX, 12, 1.0),             # R = 6
        _s("ETCH", "Mr", OX, 4, 1.0),              # R = 2, masked
        _s("DEPO", None, NIT, 6, 0.5),             # R = 3
        _s("ETCH", "Mc", NIT, 6), _s("DEPO", None, M6, 30),
        _s("ETCH", "M", M6, 30, 0.5),              # R = 15, lateral reach 15 nm (177 taps)
        _s("DEPO", None, M7, 4, 1.0),              # R = 2
        _s("ETCH", "Mr", ALL, 8),
    ]


def flow_c5():
    """32768^2: C4's pattern extended to 40 steps (slices are extracted after every step)."""
    a = flow_c4()
    b = [dict(s) for s in a[2:]]                  # second pass without the substrate/first oxide
    extra = [_s("DEPO", None, OX, 9, 1.0), _s("ETCH", "Mshift", OX, 5)]
    return a + b + extra


def flow_cw():
    """Not one of the five benchmark configurations: a patterned stack taken through the reference's
    incremental models (WETETCH undercut, ALD refill with pinch-off, STRIP, CMP) for parity at scale."""
    return [
        _s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 8), _s("DEPO", None, NIT, 12),
        _s("ETCH", "M", ALL, 26), _s("WETETCH", None, OX, 5), _s("ALD", None, M6, 3),
        _s("ETCH", "Mc", ALL, 14), _s("WETETCH", None, ALL, 2.5), _s("ALD", None, M7, 6),
        _s("CMP", None, 0, 7), _s("STRIP", None, M6, 1), _s("DEPO", None, OX, 6, 1.0),
        _s("WETETCH", None, M7, 3.3), _s("ALD", None, NIT, 2.2),
    ]


def flow_cwet():
    """Not a benchmark configuration: WETETCH undercuts between lateral steps, STRIP and CMP, for the
    parity tests of the slab driver (etch_wet across slab edges without deposit_ALD in between)."""
    return [
        _s("SUBSTR", None, SI, 30), _s("DEPO", None, OX, 8), _s("DEPO", None, NIT, 12),
        _s("ETCH", "M", ALL, 26), _s("WETETCH", None, OX, 5), _s("DEPO", None, M6, 3, 1.0),
        _s("ETCH", "Mc", ALL, 14), _s("WETETCH", None, ALL, 2.5), _s("DEPO", None, M7, 6, 1.0),
        _s("CMP", None, 0, 7), _s("STRIP", None, M6, 1), _s("DEPO", None, OX, 6, 1.0),
        _s("WETETCH", None, M7, 3.3), _s("ETCH", "Mshift", ALL, 9, 0.6), _s("WETETCH", None, NIT, 1.7),
    ]


def flow_cs():
    """Not a benchmark configuration: gaps, enclosed cavities, STRIP and CMP with and without a
    stopper, for the parity tests of the slab driver."""
    return [
        _s("SUBSTR", None, SI, 20), _s("DEPO", None, OX, 8), _s("DEPO", None, NIT, 12), _s("ETCH", "M", ALL, 25),
        _s("STRIP", None, OX, 1), _s("DEPO", None, M6, 6, 1.0), _s("ETCH", "Mc", ALL, 30), _s("STRIP", None, OX, 1),
        _s("DEPO", None, M7, 5, 1.0), _s("CMP", None, 0, 4), _s("ETCH", "Mshift", M7, 4, 1.0), _s("CMP", None, NIT, 30),
        _s("STRIP", None, M6, 1), _s("DEPO", None, OX, 7, 1.0), _s("CMP", None, SI, 40), _s("DEPO", "M", NIT, 3, 0.8),
    ]


FLOWS = {"c1": flow_c1, "c2": flow_c2, "c3": flow_c3, "c4": flow_c4, "c5": flow_c5, "cw": flow_cw, "cs": flow_cs, "cwet": flow_cwet}


def apply_step(grid, step, masks):
    """Dispatch one step on a grid with the reference's ColumnGrid method names
    (template/script_runtime.js:162-172).  masks: name -> whatever the grid takes as a mask."""
    m = None if step["mask"] is None else masks[step["mask"]]
    if step["kind"] == "SUBSTR":
        grid.deposit_general(m, step["mat"], step["t"], 0.0)
    elif step["kind"] == "DEPO":
        grid.deposit_general(m, step["mat"], step["t"], step["S"])
    elif step["kind"] == "ETCH":
        grid.etch_general(m, step["mat"], step["t"], step["S"])
    elif step["kind"] == "ALD":
        grid.deposit_ALD(step["mat"], step["t"])
    elif step["kind"] == "WETETCH":
        grid.etch_wet(step["mat"], step["t"])
    elif step["kind"] == "STRIP":
        grid.strip_connected(step["mat"])
    elif step["kind"] == "CMP":
        grid.cmp(step["t"], step["mat"])
    else:
        raise ValueError(step["kind"])


# ---------------------------------------------------------------------------------------------
# algorithmic bytes of a step (reference field sizes: len 1, mat 1, z0q 2, z1q 2, mask 1)
# ---------------------------------------------------------------------------------------------
def algorithmic_bytes(step_kind, stats, in_pass=False):
    """stats = cg_step_stats of the step: [visited, open, extended, appended, etched, removed, floor_scanned, -].

    in_pass: the step is applied in the same pass over the columns as a step before it (a fused or folded launch), so
    the column's length and top layer are at hand: it is charged its mask byte per column, what it writes (extend 2 B,
    append 6 B; etch z1 + len 3 B and 5 B per removed layer) and the deeper layers it reads (5 B per removed layer).

    deposit (vertical or conformal, fused): extend 7 B (R len, mask, top mat, top z1; W z1),
    append 11 B (R 5; W mat, z0, z1, len), any other column 2 B (mask, len).
    etch: a column losing p whole layers and ending in a partial cut 9 + 11 p B
    (R mask, len, 5 per visited layer; W 5 per removed layer, z1, len), any other column 2 B;
    isotropic etch adds 3 B per layer scanned for the floor.
    """
    n, _open, ext, app, etched, removed, floor = stats[:7]
    if in_pass:
        if step_kind in ("SUBSTR", "DEPO"):
            return n + 2 * ext + 6 * app
        return n + 3 * etched + 10 * removed + 3 * floor
    if step_kind in ("SUBSTR", "DEPO"):
        return 2 * (n - ext - app) + 7 * ext + 11 * app
    return 2 * (n - etched) + 9 * etched + 11 * removed + 3 * floor


def wet_step_bytes(n_cols, layers_before, layers_after):
    """Algorithmic bytes of one etch_wet increment (script_columngrid3.js:52-308), in the terms of algorithmic_bytes:
    every column's stack is read once (1 B len + 5 B per layer: mat, z0/z1 as the reference stores them; what the
    neighbours need of it is the same bytes again and counts as re-use) and the stack that comes out is written."""
    return 2 * n_cols + 5 * (layers_before + layers_after)


def ald_coat_bytes(n_cols, layers_before, layers_after, seeds):
    """Algorithmic bytes of one coat of deposit_ALD (computeCoatBitsSingleT + applyCoatFromBits,
    script_columngrid2.js:466-707): the stacks as in wet_step_bytes, plus every seed once (two u16 ends as the kernel
    keeps them, 8 B with the padding of the reference's Float32 pair) and the 2-byte seed count of every column."""
    return wet_step_bytes(n_cols, layers_before, layers_after) + 2 * n_cols + 8 * seeds

