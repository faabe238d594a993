#!/bin/bash
# per-dispatch HBM traffic of every grid kernel of one flow pass (rocprofv3 PMC, two runs)
TAG=${1:-steps}
SCRIPT=${2:-tools/step_times.py}     # tools/fused_times.py: the same per launch with runs of vertical steps fused
export REPS=2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $SCRIPT > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $SCRIPT > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $SCRIPT > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - <<PY
import csv, glob
def rows(pat):
    f = glob.glob("$OUT/" + pat, recursive=True)[0]
    return list(csv.DictReader(open(f)))
keep = lambda n: ("cgk::" in n) and not any(x in n for x in ("mask", "checksum", "gather", "scatter", "max_top"))
fe = [r for r in rows("pmc_fetch/**/*counter_collection.csv") if keep(r["Kernel_Name"])]
wr = [r for r in rows("pmc_write/**/*counter_collection.csv") if keep(r["Kernel_Name"])]
tr = [r for r in rows("trace/**/*kernel_trace.csv") if keep(r["Kernel_Name"])]
print("dispatches", len(fe), len(wr), len(tr), "| fetch = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB (separate rocprofv3 --pmc passes), time from --kernel-trace")
n = min(len(fe), len(wr), len(tr))
half = n // 2          # step_times.py runs the flow twice; print the second (untimed-stats-free) pass
for k in range(half, n):
    name = fe[k]["Kernel_Name"].split("(")[0].replace("void cgk::", "")[:44]
    # counters are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B)
    f = 2 * float(fe[k]["Counter_Value"]) * 1024 / 1e9; w = float(wr[k]["Counter_Value"]) * 1024 / 1e9
    ms = (int(tr[k]["End_Timestamp"]) - int(tr[k]["Start_Timestamp"])) / 1e6
    print("%3d %-44s fetch %7.3f GB  write %7.3f GB  %7.3f ms  (f+w)/t %7.1f GB/s" % (k - half, name, f, w, ms, (f + w) / ms * 1e3 if ms else 0))
PY
find $OUT -name "*.csv" -size +2M -delete
