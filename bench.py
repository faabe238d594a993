#!/usr/bin/env python3
"""Benchmark of the column-grid hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload: the 20-step three-mask flow of configuration C4 (fem_amd/flows.py) on a 16384 x 16384
column grid per GPU, resident in HBM.  One bench "step" is one process step of the flow applied
to every column (vertical deposit, masked anisotropic etch, conformal deposit or isotropic
etch); when the flow wraps, the grid is re-created (createNewGrid) inside the timed region, as
the reference's runtime does at the start of a rebuild (template/script_runtime.js:219-222).
With N > 1 every rank owns one 16384-row slab of a (16384*N) x 16384 domain (weak scaling, the
default; every slab carries the N = 1 mask pattern, so the work per GPU stays what it is at N = 1) or,
with --scaling strong, 16384/N rows of ONE 16384 x 16384 domain; the stencil steps exchange R boundary
rows with the neighbouring ranks over RCCL.  Consecutive vertical steps go out as one fused launch.

The bench checks what it timed: the checksum of the grid after the timed region must equal the
checksum an untimed pass (other kernel instantiations: the ones that keep statistics) reached at
the same step of the flow, and an untimed replay of the timed kernels up to that step.

Prints ONE JSON line (rank 0): metric = column updates per second, whole job.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--size", type=int, default=16384, help="columns per side per GPU (default 16384)")
    ap.add_argument("--flow", default="c4", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --n rows per GPU (default), strong = --n rows in all, split over the GPUs")
    ap.add_argument("--traffic", default=None, help="JSON of PMC-measured HBM bytes per launch (default: newest profiles/r*_traffic.json)")
    ap.add_argument("--cpu-n", type=int, default=2048, help="grid side of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-replicas", type=int, default=16, help="processes of the all-cores CPU baseline (0 = skip it)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--repeats", type=int, default=7,
                    help="how many times the timed region of --steps steps is run; ms_per_step is the median region")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD (torch.distributed.run, one rank per
    GPU), pass its output through and leave with its exit code.  This parent never imports torch.cuda nor touches
    the GPU, and it does not replace itself with the launcher (no exec)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def build_masks(fem, flows, grid, specs, names, nx, ny, i0, LX, LY, device, shift_x_nm=0.0):
    """Device rasters for the masks a flow uses, made by the engine's rasteriser from the fabric-style
    shape records of the reference (periodic masks: the shapes that reach this rank's rows), and checked
    against the closed form of the periodic ones.  shift_x_nm moves the shapes of the non-periodic masks
    along x (weak scaling: every slab gets its own copy of the pattern, see main)."""
    import torch
    out = {}
    for name in names:
        spec = specs[name]
        if spec["kind"] == "shapes":
            data = spec["data"]
            if shift_x_nm:
                data = {"objects": [dict(o, left=o["left"] + shift_x_nm) for o in data["objects"]]}
            out[name] = grid.mask_from_shapes(fem.reduce_mask(data, packed=True))
            continue
        data = flows.shapes_of(spec, LX, LY, x_range=(2.0 * i0, 2.0 * (i0 + nx)))
        out[name] = grid.mask_from_shapes(fem.reduce_mask(data, packed=True))
        del data
        want = flows.raster_torch(spec, nx, ny, i0, device)
        got = torch.from_numpy(out[name].download()).to(device)
        same = bool((got == want).all())
        del want, got
        if not same:
            raise RuntimeError(f"mask {name}: the rasterised shape list differs from the closed form")
    return out


def git_blob_hash(path):
    """The id git gives the file's contents (no repository needed)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def newest_traffic_file():
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")):
        m = re.match(r"r(\d+)_traffic\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def cpu_baseline(flows, flow, n):
    """The CPU restatement (oracle/, one thread) on an n x n sample of the same flow."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import OracleGrid
    import numpy as np
    specs = flows.mask_specs(2.0 * n)
    o = OracleGrid(2.0 * n, 2.0 * n, 2.0, 2.0)
    masks = {}
    for s in flow:
        m = s["mask"]
        if m and m not in masks:
            sp = specs[m]
            if sp["kind"] == "shapes":
                import fem_amd
                masks[m] = o.mask_raster(fem_amd.reduce_mask(sp["data"]))
            else:
                masks[m] = flows.raster_numpy(sp, n, n)
    t0 = time.perf_counter()
    for s in flow:
        flows.apply_step(o, s, masks)
    dt = time.perf_counter() - t0
    return {"value": n * n * len(flow) / dt / 1e9, "unit": "Gcol/s", "cores": 1, "kind": "port",
            "sample": f"{n}x{n} grid, the same {len(flow)}-step flow once, oracle/colgrid_oracle.c single thread, "
                      f"masks pre-rasterised, {dt:.1f} s"}


def cpu_baseline_all_cores(args, nproc):
    """The same CPU sample as `nproc` independent replicas, one process each (the restatement is
    single-threaded; columns of different grids are independent, so this is what all cores give)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--flow", args.flow, "--cpu-n", str(args.cpu_n)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(nproc)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    dt = time.perf_counter() - t0
    if any(p.returncode for p in procs):
        raise RuntimeError("a CPU replica failed")
    per = [json.loads(o.strip().splitlines()[-1])["value"] for o in outs]
    return {"value": sum(per), "unit": "Gcol/s", "cores": nproc, "kind": "port",
            "sample": f"{nproc} independent replicas of the {args.cpu_n}x{args.cpu_n} sample, one single-threaded process each, "
                      f"sum of their rates, {dt:.1f} s wall"}


def main():
    args = parse()
    if args.cpu_worker:                      # child of cpu_baseline_all_cores: CPU only, never touches the GPU
        from fem_amd import flows as _flows
        print(json.dumps(cpu_baseline(_flows, _flows.FLOWS[args.flow](), args.cpu_n)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    import fem_amd
    from fem_amd import flows
    from fem_amd.slabs import SlabGrid

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus would be wrong")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if os.environ.get("BENCH_ONE_GPU"):      # rehearsal of the N > 1 path on a one-GPU box (use with BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")          # nccl = RCCL; it wants one GPU per rank
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    ny = args.n
    if args.scaling == "strong" and world > 1:
        if args.n % world:
            raise SystemExit(f"--scaling strong: {args.n} rows do not split evenly over {world} ranks")
        nxl, nxg = args.n // world, args.n
    else:
        nxl, nxg = args.n, args.n * world
    LX, LY = 2.0 * nxg, 2.0 * ny
    flow = flows.FLOWS[args.flow]()
    # Weak scaling keeps the work of every GPU what it is at N = 1: the masks that are not periodic (a window, three
    # shapes) are laid out once per slab -- the N = 1 pattern repeated along x -- instead of being stretched over the
    # whole domain, which would leave some slabs fully open and others fully blocked.  Strong scaling cuts the one
    # N = 1 domain, pattern and all.
    weak_many = world > 1 and nxg != args.n
    specs = flows.mask_specs(2.0 * nxl if weak_many else LX)
    shift_x = 2.0 * rank * nxl if weak_many else 0.0
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        grid = fem_amd.ColumnGrid(dx=2.0, dy=2.0, slab=(nxg, ny, rank * nxl, nxl))
        grid.set_stream(stream.cuda_stream)
        names = sorted({s["mask"] for s in flow if s["mask"]})
        masks = build_masks(fem_amd, flows, grid, specs, names, nxl, ny, rank * nxl, LX, LY, device, shift_x)
        slab = SlabGrid(grid, rank, world, device=device, dx=2.0, dy=2.0, ny=ny, stream=stream)

        # C5 also extracts cross-sections after every step: 64 YZ lines at i = 512 s + 256 and 64 XZ lines
        # at the same j, into device buffers in the reference layout (one launch per axis)
        slice_idx = [(512 * q + 256) % min(nxl, ny) for q in range(64)] if (args.flow == "c5" and world == 1) else None

        def run(nsteps, start=0):
            k = start
            for _ in range(nsteps):
                if k % len(flow) == 0:
                    slab.createNewGrid()
                flows.apply_step(slab, flow[k % len(flow)], masks)
                if slice_idx:
                    grid.slices_device("YZ", slice_idx); grid.slices_device("XZ", slice_idx)
                k += 1
            return k

        # ---- pass 0 (untimed): per-step column counts for the algorithmic-byte model
        grid.stats_enable(True)
        step_bytes, step_bytes_in, step_check = [], [], []
        slab.createNewGrid()
        for s in flow:
            flows.apply_step(slab, s, masks)
            st = grid.stats()
            step_bytes.append(flows.algorithmic_bytes(s["kind"], st))
            step_bytes_in.append(flows.algorithmic_bytes(s["kind"], st, in_pass=True))
            step_check.append(slab.checksum())
        grid.stats_enable(False)

        # ---- warmup
        k = run(args.warmup, 0)
        grid.sync()

        # ---- timed regions: --repeats times EXACTLY --steps steps, each bracketed by a barrier and a device
        #      synchronisation on both sides; the flow simply goes on from one region into the next.  What is
        #      reported is the MEDIAN region (minimum and maximum beside it).
        reps = max(1, args.repeats)
        grid.profile(True)
        region_s = []
        kk = k
        for _ in range(reps):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kk = run(args.steps, kk)
            grid.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            region_s.append(time.perf_counter() - t0)
        prof = grid.profile(False)
        swap_s, halo_b, slab.swap_seconds, slab.halo_bytes = slab.swap_seconds, slab.halo_bytes, 0.0, 0
        total_steps = args.steps * reps

        # ---- the state the timed region left must be the state of that step of the flow
        last = (k + total_steps - 1) % len(flow)           # index in the flow of the last timed step
        check = slab.checksum()
        slab.createNewGrid()
        for s in flow[:last + 1]:
            flows.apply_step(slab, s, masks)
        replay = slab.checksum()
        if check != step_check[last] or replay != check:
            raise SystemExit("bench.py: the grid after the timed region does not match step %d of the flow: timed %016x, "
                             "untimed pass with statistics %016x, untimed replay %016x" % (last, check, step_check[last], replay))

    if world > 1:                                 # a region takes as long as its slowest rank
        t = torch.tensor(region_s, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(v) for v in t.tolist()]
    dt = sorted(region_s)[len(region_s) // 2]

    if rank == 0:
        # algorithmic bytes of the timed steps, per kernel kind (the kernel that applies the step)
        def kind_of(s):
            t = max(0.0, s["t"]); S = 0.0 if s["kind"] == "SUBSTR" else max(0.0, min(1.0, s["S"]))
            stencil = t * S >= 2.0
            if s["kind"] in ("SUBSTR", "DEPO"):
                return "k_stencil<deposit>" if stencil else "k_vertical_g4<deposit>"
            return "k_stencil<etch>" if stencil else "k_vertical_g4<etch>"
        # Consecutive vertical steps go out as ONE launch (k_vertical_fused, up to FUSE_MAX steps): the library queues
        # them until something else wants the grid -- a lateral step, createNewGrid at the wrap of the flow, the
        # cross-sections of C5, the end of the timed region.  The same rule here says which launch a step's bytes belong to.
        FUSE_MAX = 6
        fusing = not os.environ.get("CG_NO_FUSE")
        # A lateral deposit takes vertical steps along inside its own launch (k_stencil_lean): up to FUSE_MAX - 1 that follow
        # it when its radius is <= 4 (FOLD form), and, on an undivided grid and without a mask, the run that was waiting on
        # the reset grid in front of it (FRESH form).  On a slab the launch covers the rows that read no neighbour's
        # message; the rows that do get the deposit from the general kernel and the steps from fused launches of their own.
        folding = fusing and not os.environ.get("CG_NO_FOLD") and not slice_idx
        FOLDED = "k_stencil<deposit+vertical>"
        launches, cur = [], None              # (kernel group, [step indices]); cur = a launch that may still take steps: [kind, steps, room]
        # On a slab the run on the reset grid goes along with the deposit behind it when the transfers are ordered by streams
        # (RCCL); when they are staged through the host the synchronisation in front of them launches the run on its own.
        # Either way the rows that the slab's edge launches read get the run when the step is opened: one more fused launch
        # per edge band (rank 0 has one band).
        front_ok = folding and (world == 1 or slab.device_side_comm())
        front_of, edge_runs = {}, 0           # launch index -> steps of the run in front of its deposit

        def close():
            nonlocal cur
            if cur:
                kind, steps, _ = cur
                if kind == "run":
                    launches.append(("k_vertical_fused" if len(steps) >= 2 else kind_of(flow[steps[0]]), steps))
                else:
                    launches.append((FOLDED if len(steps) >= 2 else "k_stencil<deposit>", steps))
            cur = None
        for q in range(total_steps):
            idx = (k + q) % len(flow)
            kd = kind_of(flow[idx])
            if idx == 0:
                close()
            if kd.startswith("k_vertical"):
                if not fusing:
                    launches.append((kd, [idx])); continue
                if cur and cur[2] > 0:
                    cur[1].append(idx); cur[2] -= 1
                else:
                    close()
                    # (the run that starts on the reset grid goes out as one table look-up of up to 16 steps)
                    cur = ["run", [idx], (16 if idx == 0 else FUSE_MAX) - 1]
                if cur[2] == 0 or slice_idx:
                    close()
            else:
                s_ = flow[idx]
                small = math.ceil(max(0.0, s_["t"]) * max(0.0, min(1.0, s_["S"])) / 2.0) <= 4
                after_run = kd == "k_stencil<deposit>" and cur and cur[0] == "run" and cur[1][0] == 0 and s_["mask"] is None
                if folding and world > 1 and after_run:
                    edge_runs += 1
                if front_ok and after_run:
                    front_of[len(launches)] = len(cur[1])
                    cur = ["lat", cur[1] + [idx], FUSE_MAX - 1 if small else 0]
                    if cur[2] == 0:
                        close()
                    continue
                close()
                if folding and kd == "k_stencil<deposit>" and small:
                    cur = ["lat", [idx], FUSE_MAX - 1]
                else:
                    launches.append((kd, [idx]))
        close()
        # Algorithmic bytes per kernel group.  The first step of a launch is charged as on its own (it reads the column's
        # length and top layer); a step applied further down the same launch finds them in registers and is charged its
        # mask byte, what it writes and the deeper layers it has to read (flows.algorithmic_bytes(in_pass=True)).
        kbytes, kbytes_steps = {}, {}
        for grp, steps in launches:
            kbytes[grp] = kbytes.get(grp, 0) + step_bytes[steps[0]] + sum(step_bytes_in[i] for i in steps[1:])
            kbytes_steps[grp] = kbytes_steps.get(grp, 0) + sum(step_bytes[i] for i in steps)
        fused_launches = sum(1 for grp, st in launches if grp == "k_vertical_fused")
        fused_steps = sum(len(st) for grp, st in launches if grp == "k_vertical_fused")
        folded_launches = sum(1 for grp, st in launches if grp == FOLDED)
        folded_steps = sum(len(st) - 1 for grp, st in launches if grp == FOLDED)
        schedule = "fold" if folding else ("fuse" if fusing else "single")       # which steps share a launch
        # timed scopes the library opens per launch of the list above: one on an undivided grid; on a slab a lateral
        # step has two (the rows that read no message, the rows that do) and a deposit with steps inside has its own, a
        # plain deposit scope for the edge rows and a fused launch per edge band for the steps there (rank 0: one band)
        expected = {"k_vertical_fused": edge_runs} if edge_runs else {}
        for li, (grp, steps) in enumerate(launches):
            if world > 1 and grp in ("k_stencil<etch>", "k_stencil<deposit>"):
                expected[grp] = expected.get(grp, 0) + 2
            elif world > 1 and grp == FOLDED:
                behind = len(steps) - 1 - front_of.get(li, 0)
                for nm in (FOLDED, "k_stencil<deposit>") + (("k_vertical_fused",) if behind > 0 else ()):
                    expected[nm] = expected.get(nm, 0) + 1
            else:
                expected[grp] = expected.get(grp, 0) + 1
        kernels = {}
        for name, (ms, cnt) in prof.items():
            kernels[name] = {"ms_total": round(ms, 4), "launches": cnt, "ms_avg": round(ms / cnt, 4)}
            if name in kbytes and ms > 0:
                kernels[name]["algorithmic_GB"] = round(kbytes[name] / 1e9, 3)
                kernels[name]["GBps"] = round(kbytes[name] / 1e9 / (ms / 1e3), 1)
                expect = expected.get(name, 0)
                if expect != cnt:                           # the rule above and the library disagree on what went out together
                    kernels[name]["launches_by_bench_rule"] = expect
                if kbytes_steps[name] != kbytes[name]:      # what the same steps move when every one is a pass of its own
                    kernels[name]["algorithmic_GB_step_by_step"] = round(kbytes_steps[name] / 1e9, 3)
        # the roofline object is about the group of launches that takes the most time
        MULTI = ("k_vertical_fused", FOLDED)          # groups whose launches apply several steps
        cand = {n_: v for n_, v in kernels.items() if "GBps" in v}
        dom = max(cand, key=lambda n_: cand[n_]["ms_total"]) if cand else None
        pick_later = dom in MULTI                  # (settled below, once the traffic file has been looked at)
        roofline = None
        # HBM traffic per launch from the PMC passes of the same command (tools/prof.sh), when committed: the file holds
        # fetch + write bytes per launch of every kernel group; those and the live launch times give the rate at which
        # each group really moves bytes (traffic_GBps, traffic_frac of 8 TB/s) -- for a fused launch THIS is the
        # memory rate, not its algorithmic figure
        traffic, traffic_src, flow_traffic = None, None, None
        tpath = args.traffic or newest_traffic_file()
        if dom and tpath and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            # (the bytes of a group's launch depend on which steps go out together: same queueing rules only)
            same_workload = (tj.get("n") == args.n and tj.get("flow") == args.flow and tj.get("n_gpus", 1) == world
                             and tj.get("schedule", schedule) == schedule)
            if same_workload:
                tk = tj.get("kernels", {})
                for name, v in kernels.items():
                    if name in tk and v["ms_total"] > 0:
                        b = tk[name]["fetch_bytes_per_launch"] + tk[name]["write_bytes_per_launch"]
                        v["traffic_bytes_per_launch"] = b
                        v["traffic_GBps"] = round(b * v["launches"] / 1e9 / (v["ms_total"] / 1e3), 1)
                        v["traffic_frac"] = round(v["traffic_GBps"] / 8000.0, 4)
                timed = [n_ for n_ in kernels if n_ in kbytes]
                if timed and all("traffic_bytes_per_launch" in kernels[n_] for n_ in timed):
                    flow_traffic = sum(kernels[n_]["traffic_bytes_per_launch"] * kernels[n_]["launches"] for n_ in timed)
                if dom in tk:
                    traffic = kernels[dom]["traffic_bytes_per_launch"]
                    traffic_src = {"file": os.path.relpath(tpath, ROOT), "git_blob": git_blob_hash(tpath)}
        if pick_later and cand[dom].get("traffic_GBps") is None:
            # no measured traffic for this workload: the group of single-step launches that takes the most time, if there is one
            single = {n_: v for n_, v in cand.items() if n_ not in MULTI}
            if single:
                dom = max(single, key=lambda n_: single[n_]["ms_total"])
        if dom:
            achieved = cand[dom]["GBps"]
            note = None
            if dom in MULTI and cand[dom].get("traffic_GBps") is not None:
                # several steps per launch: no per-step byte count describes the launch (intermediate states never reach
                # memory), so the rate given is the one at which it really moves bytes
                achieved = cand[dom]["traffic_GBps"]
                note = ("a launch of this group applies several process steps in one pass over the columns; achieved = HBM bytes per launch "
                        "(rocprofv3 PMC, traffic_source) / launch time.  kernels[...].GBps is the algorithmic figure: the first step of a launch "
                        "charged in full, each further one its mask byte, what it writes and the deeper layers it reads")
            elif dom in MULTI:
                # nothing but multi-step launches ran and no traffic file covers the workload (C2 at a size that stays in
                # the caches): the algorithmic figure of such launches is not a memory rate, so none is given
                achieved = None
                note = ("every launch applies a run of process steps in one pass over the columns and no PMC traffic file covers this workload: "
                        "no memory rate is given (kernels[...].GBps charges the first step of a launch in full and each further one its "
                        "mask byte, what it writes and the deeper layers it reads)")
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4) if achieved is not None else None, "note": note, "traffic": traffic, "traffic_source": traffic_src,
                        "launches": cand[dom]["launches"], "ms_avg": cand[dom]["ms_avg"], "steps_per_launch": round(sum(len(st) for grp, st in launches if grp == dom) / max(1, sum(1 for grp, st in launches if grp == dom)), 2),
                        "algorithmic_bytes_per_launch": int(kbytes[dom] / max(1, cand[dom]["launches"]))}
        total_alg = sum(step_bytes[(k + q) % len(flow)] for q in range(total_steps))
        total_s = sum(region_s)
        cols = nxl * ny * world * args.steps
        out = {
            "metric": "column-updates/sec per process step", "value": round(cols / dt / 1e9, 4), "unit": "Gcol/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "u16 heights, f64/f32 arithmetic",
            "data": "synthetic", "flow": args.flow, "n": args.n,
            "config": {"workload": f"{args.flow.upper()} flow ({len(flow)} process steps, 3 masks{', 128 cross-sections after every step' if slice_idx else ''}) on a {nxl}x{ny} column grid per GPU",
                       "grid": [nxg, ny], "rows_per_gpu": nxl, "lmax": grid.Lmax, "parallelism": f"slab{world}",
                       "masks": "the N=1 masks, the non-periodic ones repeated once per slab" if weak_many else "as N=1",
                       "world": dist.get_world_size() if world > 1 else 1},
            "halo": {"bytes_sent_per_rank": halo_b, "bytes_per_step": round(halo_b / max(1, total_steps), 1),
                     "exchange_ms_total": round(swap_s * 1e3, 3)},
            "timing": {"regions": reps, "steps_per_region": args.steps, "ms_per_step_median": round(dt * 1e3 / args.steps, 4),
                       "ms_per_step_min": round(min(region_s) * 1e3 / args.steps, 4),
                       "ms_per_step_max": round(max(region_s) * 1e3 / args.steps, 4),
                       "note": "value and ms_per_step are the median region; kernel times below are sums over all regions"},
            "whole_flow_algorithmic_GBps": round(total_alg * world / 1e9 / total_s, 1),
            "whole_flow_traffic_GBps": round(flow_traffic * world / 1e9 / total_s, 1) if flow_traffic else None,
            "fusion": {"vertical_steps_fused": fused_steps, "fused_launches": fused_launches,
                       "vertical_steps_inside_lateral_deposits": folded_steps, "lateral_deposits_with_steps_inside": folded_launches,
                       "note": "runs of consecutive vertical steps are applied in one pass over the columns: in a launch of their own "
                               "(k_vertical_fused) or inside the lateral deposit in front of or behind them (k_stencil<deposit+vertical>); "
                               "CG_NO_FOLD=1 keeps them out of the lateral launches, CG_NO_FUSE=1 launches every step on its own"},
            "schedule": schedule,
            "checksum": "%016x" % check,
            "roofline": roofline, "kernels": kernels,
        }
        if not args.no_cpu and world == 1:          # the CPU baseline is timed at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(flows, flow, args.cpu_n)
            except Exception as e:  # the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "Gcol/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
            if args.cpu_replicas > 0 and world == 1:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, args.cpu_replicas)
                except Exception as e:
                    out["cpu_baseline_all_cores"] = {"value": None, "unit": "Gcol/s", "cores": args.cpu_replicas, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    grid.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
