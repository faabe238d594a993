"""Wider seeded sweeps of the GPU-vs-oracle parity tests (not collected by pytest; run on a GPU box as
`python tests/stress_gpu.py`): the random-flow tests of test_gpu_parity.py with 88 more seeds and sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_parity as t          # noqa: E402

for seed in range(5, 25):
    nx, ny = 96 + 4 * ((seed * 37) % 60), 64 + 4 * ((seed * 53) % 70)
    if seed % 5 == 0:
        nx += 1                      # the one-column kernels
    t.test_random_flow_vs_oracle(nx, ny, seed, stats=True)
    t.test_random_flow_vs_oracle(nx, ny, seed, stats=False)        # the production kernel instantiations
    print("general seed", seed, nx, ny, "ok", flush=True)
for n in (50, 400, 1200):
    for slab in (None, (70, 150)):
        t.test_mask_rasteriser_many_shapes(n, slab)
for seed in range(30, 46):
    nx, ny = 64 + 4 * ((seed * 29) % 30), 60 + 4 * ((seed * 41) % 30)
    t.test_random_flow_with_wet_etch_and_ald_vs_oracle(nx, ny, seed)
    print("wet/ald seed", seed, nx, ny, "ok", flush=True)
for seed in range(50, 62):
    nx, ny = 80 + 4 * ((seed * 29) % 50), 72 + 4 * ((seed * 41) % 50)
    t.test_random_flow_with_strip_cmp_cavity_vs_oracle(nx, ny, seed)
    print("strip/cmp seed", seed, nx, ny, "ok", flush=True)
for seed in range(70, 110):
    nx, ny = 32 + 4 * ((seed * 29) % 40), 36 + 4 * ((seed * 41) % 60)
    t._check_runs_of_vertical_steps(nx, ny, seed, [0.1, 0.05, 0.2, 0.1][seed % 4])
    print("fused runs seed", seed, nx, ny, "ok", flush=True)
for seed in range(120, 136):
    # lateral deposits that wait for the steps behind them (and take the run on a reset grid along): undivided grids and
    # groups of 2..4 slabs whose slabs have inner rows (tiles are 64 rows) and whole workgroups of columns per edge band
    parts = seed % 4 + 1
    nx, ny = parts * (192 + 64 * (seed % 3)), [128, 256, 512, 1024][(seed // 4) % 4]
    t._check_steps_inside_lateral_deposits(nx, ny, seed, 0 if parts == 1 else parts)
    print("steps inside lateral deposits seed", seed, nx, ny, parts, "ok", flush=True)
print("stress ok")
