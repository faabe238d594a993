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
