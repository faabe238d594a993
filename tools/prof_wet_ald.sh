#!/bin/bash
# kernel-trace summary (8192^2) and SQ counters (4096^2) of the etch_wet / deposit_ALD sequence of tools/time_wet_ald.py
TAG=${1:-wetald}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_wet_ald.py 8192 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
{
echo "# python3 tools/time_wet_ald.py 8192 under rocprofv3 --kernel-trace --stats (wall times printed by the script include the tracer)"
grep -E "^etch_wet|^deposit_ALD" $OUT/trace.log
echo
echo "# kernels of the sequence: total ms, launches, average ms"
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict(); on = False
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cgk::", "").replace("cgk::", "")
    if "k_wet_step" in n: on = True
    if not on or not n.startswith("k_"): continue
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %9.2f ms %5d launches %8.3f ms avg" % (n[:28], ms, c, ms / c))
print()
print("# roofline (HBM, 8000 GB/s): algorithmic bytes of fem_amd/flows.py (wet_step_bytes, ald_coat_bytes) over the launch times above")
for line in open("$OUT/trace.log"):
    if line.startswith("MODEL "):
        _, k, _, cnt, _, b = line.split()
        hit = [v for n_, v in agg.items() if n_.startswith(k)]        # (k_ald_coat: k_ald_coat_runs or k_ald_coat itself)
        c, ms = sum(v[0] for v in hit), sum(v[1] for v in hit)
        if c == int(cnt) and ms > 0:
            r = int(b) / 1e9 / (ms / 1e3)
            print('{"kernel": "%s", "bound": "hbm", "launches": %d, "algorithmic_bytes_per_launch": %d, "ms_avg": %.3f, "achieved": %.1f, "peak": 8000.0, "unit": "GB/s", "frac": %.4f}' % (k, c, int(b) // c, ms / c, r, r / 8000.0))
        else:
            print("# %s: %d launches modelled, %d traced" % (k, int(cnt), c))
PY
echo
echo "# SQ counters per wave (python3 tools/time_wet_ald.py 4096, rocprofv3 --pmc, one column per lane and loop trip)"
bash tools/prof_sq_ald.sh ${TAG}_sq 4096
} > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
