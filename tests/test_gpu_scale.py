"""Benchmark flows at scale: oracle equality where the oracle finishes in seconds, and
size-independent properties at the full 16384 x 16384 benchmark size."""
import numpy as np
import pytest

from oracle import OracleGrid

pytestmark = pytest.mark.gpu


def _masks(fem, flows, g, o, specs, names, n):
    import torch
    gm, om = {}, {}
    for name in names:
        spec = specs[name]
        if spec["kind"] == "shapes":
            red = fem.reduce_mask(spec["data"])
            gm[name] = g.mask_from_shapes(red)
            if o is not None:
                om[name] = o.mask_raster(red)
        else:
            r = flows.raster_torch(spec, n, n, 0, torch.device("cuda", 0))
            torch.cuda.synchronize()
            gm[name] = g.mask_from_device_raster(r.data_ptr(), r.numel())
            g.sync()
            if o is not None:
                om[name] = flows.raster_numpy(spec, n, n)
                # closed form == the reference predicate on the expanded shape list (small sizes only)
                if n <= 512:
                    np.testing.assert_array_equal(om[name], o.mask_raster(fem.reduce_mask(flows.shapes_of(spec, 2 * n, 2 * n))))
                np.testing.assert_array_equal(gm[name].download(), om[name])
    return gm, om


@pytest.mark.parametrize("flow_name,n,stats", [("c2", 512, True), ("c3", 512, True), ("c4", 1024, True), ("c5", 384, True), ("cw", 512, True),
                                               ("c2", 512, False), ("c3", 768, False), ("c5", 512, False)])
def test_benchmark_flows_vs_oracle(flow_name, n, stats):
    import fem_amd as fem
    from fem_amd import flows
    flow = flows.FLOWS[flow_name]()
    specs = flows.mask_specs(2.0 * n)
    g = fem.ColumnGrid(2.0 * n, 2.0 * n, 2.0, 2.0)
    o = OracleGrid(2.0 * n, 2.0 * n, 2.0, 2.0)
    gm, om = _masks(fem, flows, g, o, specs, sorted({s["mask"] for s in flow if s["mask"]}), n)
    g.stats_enable(stats)      # off: the instantiations a production run uses (no case counts to compare then)
    g.keep_step_snapshots = False
    for k, s in enumerate(flow):
        flows.apply_step(g, s, gm)
        flows.apply_step(o, s, om)
        o.slider_reset()
        assert g.checksum() == o.checksum(), f"{flow_name} step {k} {s}"
        if stats and s["kind"] in ("SUBSTR", "DEPO", "ETCH"):
            gs, os_ = g.stats(), o.stats()
            assert gs[2:6] == os_[2:6], f"{flow_name} step {k} column counts {gs} vs {os_}"
        if flow_name == "c5":     # C5 extracts cross-sections after every step
            idx = [min(n - 1, 37 * q + 5) for q in range(8)]
            ln, mat, zp = o.arrays(); L = o.Lmax
            for axis in ("YZ", "XZ"):
                sl, sm, sz = g.slices(axis, idx)
                for q, i in enumerate(idx):
                    ref_len = ln.reshape(n, n)[i] if axis == "YZ" else ln.reshape(n, n)[:, i]
                    ref_mat = mat.reshape(n, n, L)[i] if axis == "YZ" else mat.reshape(n, n, L)[:, i]
                    np.testing.assert_array_equal(sl[q], ref_len)
                    np.testing.assert_array_equal(sm[q][:, :L], ref_mat[:, :g.Lmax])
    assert g.Lmax == o.Lmax
    ln, mat, zp = g.download()
    oln, omat, ozp = o.arrays()
    np.testing.assert_array_equal(ln, oln); np.testing.assert_array_equal(mat, omat); np.testing.assert_array_equal(zp, ozp)
    g.close()


def test_bench_call_path_vs_oracle():
    """The benchmark drives the grid through SlabGrid(world=1) -> cg_stencil_begin / cg_stencil_finish (not
    cg_deposit_general) and re-creates the grid when the flow wraps: the same calls here, against the oracle."""
    import fem_amd as fem
    from fem_amd import flows
    from fem_amd.slabs import SlabGrid
    n = 1024
    flow = flows.flow_c4()
    specs = flows.mask_specs(2.0 * n)
    g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
    o = OracleGrid(2.0 * n, 2.0 * n, 2.0, 2.0)
    gm, om = _masks(fem, flows, g, o, specs, sorted({s["mask"] for s in flow if s["mask"]}), n)
    slab = SlabGrid(g, 0, 1, dx=2.0, dy=2.0, ny=n)
    want = []
    for s in flow:
        flows.apply_step(o, s, om)
        want.append(o.checksum())
    for rep in range(2):                       # the second pass starts from createNewGrid on a used handle, as the bench's wrap does
        slab.createNewGrid()
        for k, s in enumerate(flow):
            flows.apply_step(slab, s, gm)
            assert slab.checksum() == want[k], f"pass {rep} step {k} {s}"
    # without a look at the grid in between (the timed region never reads anything back)
    slab.createNewGrid()
    for s in flow:
        flows.apply_step(slab, s, gm)
    assert slab.checksum() == want[-1]
    g.close()


def test_full_size_properties():
    """16384 x 16384 (the benchmark size): no oracle, only properties that do not depend on size."""
    import fem_amd as fem
    from fem_amd import flows
    n = 16384
    specs = flows.mask_specs(2.0 * n)
    g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
    gm, _ = _masks(fem, flows, g, None, specs, ["M", "Mc", "Mr"], n)
    g.deposit_general(None, 3, 30, 0)
    c0 = g.checksum()
    assert g.maxHeight() == 30.0
    # deposit + equal etch of the same material is the identity (vertical and lateral)
    g.deposit_general(None, 4, 20, 0); g.etch_general(None, 4, 20, 0)
    assert g.checksum() == c0
    g.deposit_general(None, 4, 6, 1.0); g.etch_general(None, 4, 6, 0)
    assert g.checksum() == c0
    # a masked etch followed by the etch of the complement equals the unmasked etch
    g.snapshot_save(0)
    g.etch_general(gm["M"], 3, 7.5, 0)
    inv = g.mask_from_raster(1 - gm["M"].download())
    g.etch_general(inv, 3, 7.5, 0)
    c_split = g.checksum()
    g.snapshot_load(0)
    g.etch_general(None, 3, 7.5, 0)
    assert g.checksum() == c_split and g.maxHeight() == 22.5
    # translation symmetry: the periodic mask Mc has pitch 24 cells; a conformal step on a periodic
    # state keeps row i and row i + 24*k identical away from the domain edges
    g.snapshot_load(0)
    g.etch_general(gm["Mc"], 3, 12, 0)
    g.deposit_general(None, 5, 10, 1.0)
    g.etch_general(None, 5, 6, 1.0)
    sl, sm, sz = g.slices("YZ", [48, 48 + 24 * 100, 48 + 24 * 600])
    np.testing.assert_array_equal(sl[0], sl[1]); np.testing.assert_array_equal(sz[0], sz[2]); np.testing.assert_array_equal(sm[0], sm[1])
    # sum of slab checksums == whole grid (the checksum is a sum over columns keyed by global index)
    g.close()


def test_plane_growth_inside_a_conformal_step():
    """Lmax 2 -> 18: the overflow re-run of vertical and conformal deposits against the oracle.

    The flow stops before a column is exactly Lmax layers tall at the end of a conformal step: there the
    reference's identify_cavity writes its top cavity piece past the column's slots into the next
    column and leaves len = Lmax + 1 (script_columngrid2.js:60-69 has no bound check), after which its
    state is corrupt.  The engine does not reproduce that."""
    import fem_amd as fem
    n = 96
    g = fem.ColumnGrid(2.0 * n, 2.0 * n, 2.0, 2.0, 2)
    o = OracleGrid(2.0 * n, 2.0 * n, 2.0, 2.0, lmax=2)
    rng = np.random.RandomState(5)
    m = (rng.rand(n * n) < 0.6).astype(np.uint8)
    gmask = g.mask_from_raster(m)
    steps = [("d", None, 3, 10, 0), ("d", gmask, 4, 5, 0), ("d", None, 5, 6, 1.0), ("d", gmask, 6, 4, 0.8), ("e", gmask, 255, 7, 1.0)]
    steps += [("d", None if k % 2 else gmask, 3 + k % 5, 2 + k % 3, (k % 3) / 2) for k in range(12)]
    for idx, (kind, mk, mat, t, S) in enumerate(steps):
        om = None if mk is None else m
        if kind == "d":
            g.deposit_general(mk, mat, t, S); o.deposit_general(om, mat, t, S)
        else:
            g.etch_general(mk, mat, t, S); o.etch_general(om, mat, t, S)
        assert g.Lmax == o.Lmax, f"step {idx}"
        assert g.checksum() == o.checksum(), f"step {idx}"
    ln, mat, zp = g.download(); oln, omat, ozp = o.arrays()
    np.testing.assert_array_equal(ln, oln); np.testing.assert_array_equal(mat, omat); np.testing.assert_array_equal(zp, ozp)
    g.close()


def test_edge_cases():
    import fem_amd as fem
    # 1 x 1 grid, empty grid ops, zero / negative thickness, mask with no open column
    g = fem.ColumnGrid(2, 2, 2, 2)
    assert (g.NX, g.NY) == (1, 1)
    assert g.maxHeight() == 1.0                       # empty grid: maxHeight() is 1 (script_columngrid2.js:103)
    g.etch_general(None, 255, 5, 0); g.etch_general(None, 255, 5, 1.0)
    g.deposit_general(None, 3, 0, 0); g.deposit_general(None, 3, -4, 1.0)
    assert g.download()[0][0] == 0
    g.deposit_general(None, 3, 6553.5, 0)
    g.deposit_general(None, 3, 0.2, 0)               # Uint16 wrap: 65537 -> 1
    o = OracleGrid(2, 2, 2, 2); o.deposit_general(None, 3, 6553.5, 0); o.deposit_general(None, 3, 0.2, 0)
    assert g.checksum() == o.checksum()
    g.close()
    g = fem.ColumnGrid(14, 22, 2, 2)                 # 7 x 11: no vector path
    o = OracleGrid(14, 22, 2, 2)
    closed = g.mask_from_raster(np.zeros(77, np.uint8))
    for args in [(None, 3, 10, 0), (closed, 4, 5, 0), (closed, 4, 5, 1.0), (None, 4, 3, 1.0)]:
        g.deposit_general(*args); o.deposit_general(None if args[0] is None else np.zeros(77, np.uint8), *args[1:])
    g.etch_general(closed, 255, 5, 1.0); o.etch_general(np.zeros(77, np.uint8), 255, 5, 1.0)
    assert g.checksum() == o.checksum()
    with pytest.raises(fem.ColGridError):
        g.deposit_general(fem.Mask(g, 99), 3, 1, 0)  # unknown mask id
    g.close()
    with pytest.raises(fem.ColGridError):
        fem.ColumnGrid(10, 10, 0, 2)                 # bad pitch
    g = fem.ColumnGrid(20, 20, 4, 2)                 # dy < dx: a lateral step that reaches further along y than along x is refused
    g.deposit_general(None, 3, 10, 0)
    with pytest.raises(fem.ColGridError):
        g.deposit_general(None, 4, 10, 1.0)          # Rx = 3, Ry = 5
    g.close()


def test_c5_first_steps_at_32768():
    """32768 x 32768 (configuration C5's size, 1.07 G columns): the first 12 steps of its flow -- vertical runs that go out
    fused, lateral steps on the lean streaming kernels -- against a replay of the same steps through the other kernel
    family (statistics on: one launch per step, gathering lateral kernels, the reference's double arithmetic throughout):
    same checksum after the last step, same cross-sections.  The three masks of these steps are periodic (32, 24 and 32
    cells), so away from the domain edges the grid repeats every 96 cells in both directions."""
    import torch
    import fem_amd as fem
    from fem_amd import flows
    n = 32768
    free, _total = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs 120 GiB of free device memory")
    flow = flows.flow_c5()[:12]
    assert {s["mask"] for s in flow if s["mask"]} == {"M", "Mc", "Mshift"}
    specs = flows.mask_specs(2.0 * n)
    g = fem.ColumnGrid(dx=2.0, dy=2.0, slab=(n, n, 0, n))
    g.keep_step_snapshots = False
    gm, _ = _masks(fem, flows, g, None, specs, ["M", "Mc", "Mshift"], n)
    lines_i = [4800, 4800 + 96 * 100, 4800 + 96 * 250, n - 1]
    lines_j = [4801, 4801 + 96 * 7, 0]

    def run(stats):
        g.createNewGrid()
        g.stats_enable(stats)
        for s in flow:
            flows.apply_step(g, s, gm)
        out = (g.checksum(), g.maxHeight(), g.Lmax, g.slices("YZ", lines_i), g.slices("XZ", lines_j))
        g.stats_enable(False)
        return out

    a = run(False)
    b = run(True)
    assert a[:3] == b[:3]
    for x, y in zip(a[3] + a[4], b[3] + b[4]):
        np.testing.assert_array_equal(x, y)
    # translation symmetry (period 96 cells), away from the edges
    ln, mat, zp = a[3]
    np.testing.assert_array_equal(ln[0], ln[1]); np.testing.assert_array_equal(zp[0], zp[2]); np.testing.assert_array_equal(mat[1], mat[2])
    inner = slice(960, 960 + 96 * 300)
    np.testing.assert_array_equal(ln[0][inner], ln[0][960 + 96:960 + 96 * 301])
    ln2, mat2, zp2 = a[4]
    np.testing.assert_array_equal(ln2[0][inner], ln2[1][inner]); np.testing.assert_array_equal(zp2[0][inner], zp2[1][inner])
    g.close()
