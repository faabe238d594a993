#!/bin/bash
# One rocprofv3 --pmc pass over one flow pass (tools/step_times.py, REPS=1); prints the counters of
# every lateral / vertical step kernel dispatch.  usage: tools/pmc_kernels.sh <tag> <counter> [<counter> ...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPS=1 NOSTATS=1 timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc -- python3 tools/step_times.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "cgk::k_stencil" not in n and "cgk::k_vertical" not in n: continue
    key = (int(r["Dispatch_Id"]), n.split("(")[0].replace("void cgk::", "")[:40])
    disp.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for d in disp.values() for c in d})
print("%-4s %-40s " % ("id", "kernel") + " ".join("%16s" % c[-16:] for c in names))
for (i, n), d in disp.items():
    print("%-4d %-40s " % (i, n) + " ".join("%16.4g" % d.get(c, 0) for c in names))
PY
find $OUT -name "*.csv" -size +2M -delete
