"""The C ABI: every function include/colgrid.h declares is exported by libcolgrid.so and bound
by the Python view (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "colgrid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from fem_amd import colgrid
    names = _declared()
    assert len(names) >= 35
    lib = ctypes.CDLL(colgrid.LIB_PATH) if False else colgrid.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in colgrid.h but not exported"
        assert n in colgrid.ABI, f"{n} has no ctypes signature in fem_amd/colgrid.py"
    for n in colgrid.ABI:
        assert n in names, f"{n} is bound but not declared in colgrid.h"
    assert lib.cg_version().startswith(b"colgrid")


def test_struct_sizes_match_the_header():
    from fem_amd import colgrid
    assert ctypes.sizeof(colgrid.CgShape) == 104       # 6 x int32 + 10 x double
    assert ctypes.sizeof(colgrid.CgInfo) == 6 * 4 + 7 * 8


def test_missing_library_fails_loudly(tmp_path):
    from fem_amd import colgrid
    import pytest
    with pytest.raises(colgrid.ColGridError):
        colgrid.load_library(str(tmp_path / "nope.so"))


def test_reduce_mask_matches_js_reduction():
    """Python and JavaScript hosts reduce a mask record to the same numbers (cos/sin bits included)."""
    import json
    from fem_amd import reduce_mask
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_case
    for name in ["masks_small", "stencil_odd", "vertical_small"]:
        c = load_case(name)
        for m in c["mask_order"]:
            r = reduce_mask(c["masks"][m])
            assert r["pts"] == c["reduced"][m]["pts"]
            assert r["shapes"] == c["reduced"][m]["shapes"], (name, m)
