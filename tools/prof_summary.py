"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    n = name.split("(")[0]
    for pre in ("void cgk::", "cgk::", "void "):
        if n.startswith(pre):
            n = n[len(pre):]
    return n[:60]


def find(pattern):
    return glob.glob(os.path.join(out, pattern), recursive=True)


# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-62s calls %6s  total %12s ns  avg %12s ns  %6s %%" % (short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))

# counters: average per dispatch per kernel
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in find("pmc_*/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
print("== counters, average per dispatch")
for k in sorted(agg):
    print("  " + k)
    for c in sorted(agg[k]):
        print("      %-22s %18.1f   (n=%d)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))

# HBM traffic per launch by the labels bench.py uses (FETCH_SIZE / WRITE_SIZE count kilobytes)
import json
import re


def parse(k):
    """-> (bench.py's group label, keeps statistics) of a step kernel, or None."""
    if k.startswith("k_vertical_fused") or k.startswith("k_vertical_fresh"):  # a run of vertical steps in one launch: never keeps statistics
        return "k_vertical_fused", False
    m = re.match(r"k_(stencil_stream|stencil_lean|stencil_sym|stencil_apply|vertical_g4|vertical_1)<(.*)>", k)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    kind = m.group(1)
    if kind in ("stencil_stream", "stencil_lean"):   # <R, ETCH, ...>: never keep statistics
        # <R, ETCH, MASKED, NF, FRESH>: NF > 0 / FRESH = the vertical steps behind / in front of a deposit inside its launch
        if kind == "stencil_lean" and (args[3] not in ("0", "false") or (len(args) > 4 and args[4] == "true")):
            return "k_stencil<deposit+vertical>", False
        return "k_stencil<%s>" % ("etch" if args[1] == "true" else "deposit"), False
    if kind == "stencil_sym":             # <R, TJS, ETCH, STATS>
        return "k_stencil<%s>" % ("etch" if args[2] == "true" else "deposit"), args[3] == "true"
    if kind == "stencil_apply":           # <ETCH, STATS>
        return "k_stencil<%s>" % ("etch" if args[0] == "true" else "deposit"), args[1] == "true"
    return "k_vertical_g4<%s>" % ("etch" if args[0] == "true" else "deposit"), args[1] == "true"


def label(k):
    p = parse(k)
    return p[0] if p else None


# launch durations grouped the way bench.py groups them (its roofline.ms_avg is the average over the
# timed launches of one group; the instances that collect statistics run in the untimed pass only)
grp = defaultdict(lambda: [0.0, 0])
for f in find("trace/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        pk = parse(k)
        if not pk or pk[1]:
            continue
        lb = pk[0]
        grp[lb][0] += float(row["TotalDurationNs"]); grp[lb][1] += int(row["Calls"])
if grp:
    print("== launch durations by bench.py's kernel groups (instances without statistics, warm-up + timed steps)")
    for lb, (ns, n) in sorted(grp.items()):
        print("  %-28s launches %4d  avg %10.4f ms  total %10.3f ms" % (lb, n, ns / n / 1e6, ns / 1e6))

tr = defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0, "nf": 0, "nw": 0})
for k in agg:
    pk = parse(k)
    if not pk or pk[1]:                   # the instances that keep statistics run in the untimed pass only
        continue
    lb = pk[0]
    if "FETCH_SIZE" in agg[k]:
        tr[lb]["fetch_kb"] += agg[k]["FETCH_SIZE"]; tr[lb]["nf"] += cnt[k]["FETCH_SIZE"]
    if "WRITE_SIZE" in agg[k]:
        tr[lb]["write_kb"] += agg[k]["WRITE_SIZE"]; tr[lb]["nw"] += cnt[k]["WRITE_SIZE"]
bench_line = {}
for f in find("trace.log"):
    for line in open(f):
        if line.startswith("{") and '"metric"' in line:
            bench_line = json.loads(line)
res = {}
for lb, v in tr.items():
    if v["nf"] and v["nw"]:
        res[lb] = {"fetch_bytes_per_launch": round(2 * v["fetch_kb"] * 1024 / v["nf"]), "write_bytes_per_launch": round(v["write_kb"] * 1024 / v["nw"]),
                   "launches_profiled": min(v["nf"], v["nw"])}
if res:
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump({"n": bench_line.get("n"), "flow": bench_line.get("flow"), "n_gpus": bench_line.get("n_gpus", 1),
                   "schedule": bench_line.get("schedule"),
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python3 bench.py --no-cpu; counters are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B; checked on the vertical deposits, which read the 4-byte top of every column: 1.074 GB expected at 16384^2)",
                   "kernels": res}, f, indent=1)
    print("== HBM traffic per launch (bytes)")
    for lb, v in sorted(res.items()):
        print("  %-28s fetch %14d  write %14d  (n=%d)" % (lb, v["fetch_bytes_per_launch"], v["write_bytes_per_launch"], v["launches_profiled"]))
