"""cg_slices_noalloc (the entry point that allocates nothing) against cg_slices, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(fem):
    from fem_amd import flows
    nx, ny = 96, 72
    g = fem.ColumnGrid(2.0 * nx, 2.0 * ny, 2.0, 2.0)
    specs = flows.mask_specs(2.0 * nx)
    flow = flows.flow_c4()[:12]
    masks = {n: g.mask_from_shapes(fem.reduce_mask(flows.shapes_of(specs[n], 2 * nx, 2 * ny))) for n in sorted({s["mask"] for s in flow if s["mask"]})}
    for s in flow:
        flows.apply_step(g, s, masks)
    return g, nx, ny


def _compare(g, axis, idx, torch):
    inf = g.info()
    line = inf.ny if axis == "YZ" else inf.nx
    L, cnt = inf.lmax, len(idx)
    dev = torch.device("cuda", 0)
    # outputs larger than needed and pre-filled: the call must write exactly its own part
    d_mat = torch.full((cnt * line * L + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_zp = torch.full((2 * cnt * line * L + 64,), 0x7777, dtype=torch.int16, device=dev)
    d_len = torch.full((cnt * line + 64,), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    g.slices_noalloc(axis, idx, d_mat.data_ptr(), d_zp.data_ptr(), d_len.data_ptr())
    g.sync()
    ln, mat, zp = g.slices(axis, idx)                       # the existing entry point
    np.testing.assert_array_equal(d_len.cpu().numpy()[:cnt * line], ln.reshape(-1))
    np.testing.assert_array_equal(d_mat.cpu().numpy()[:cnt * line * L], mat.reshape(-1))
    np.testing.assert_array_equal(d_zp.cpu().numpy().view(np.uint16)[:2 * cnt * line * L], zp.reshape(-1))
    assert (d_len.cpu().numpy()[cnt * line:] == 0xEE).all() and (d_mat.cpu().numpy()[cnt * line * L:] == 0xEE).all()


def test_noalloc_slices_match_cg_slices_twice_in_a_row():
    import torch
    import fem_amd as fem
    g, nx, ny = _build(fem)
    _compare(g, "YZ", [0, 5, 17, 40, nx - 1, 33, 2, 64], torch)     # 8 lines
    _compare(g, "YZ", [nx - 2, 7, 50], torch)                       # then fewer: the array still holds 5 stale entries
    _compare(g, "XZ", [ny - 1, 0, 31, 8], torch)
    _compare(g, "XZ", [3], torch)
    # a second handle on the same GPU shares the array: used one after the other, both are right
    g2, _, _ = _build(fem)
    g2.etch_general(None, 255, 9, 0)
    _compare(g2, "YZ", [1, 2], torch)
    _compare(g, "YZ", [9, 8, 7, 6, 5], torch)
    g2.close()
    # beyond the fixed maximum: refused, nothing written
    with pytest.raises(fem.ColGridError) as e:
        g.slices_noalloc("YZ", [0] * 4097, 1, 1, 1)
    assert "at most 4096" in str(e.value)
    g.close()
