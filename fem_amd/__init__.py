"""fem_amd -- MI355X-native engine for the column-grid process emulator.

The product is libcolgrid (HIP kernels behind the C ABI of include/colgrid.h).  This package
is the Python view of that ABI, used by the tests, the benchmark and multi-GPU slab driver;
the JavaScript view (an N-API addon with the reference's ColumnGrid / ProcessRuntime
surface) lives in fem_amd/js.
"""
from .colgrid import ColumnGrid, GroupGrid, Mask, ColGridError, load_library, reduce_mask  # noqa: F401

__all__ = ["ColumnGrid", "GroupGrid", "Mask", "ColGridError", "load_library", "reduce_mask"]
