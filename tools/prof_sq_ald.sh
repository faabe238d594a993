#!/bin/bash
# per-dispatch SQ counters of the etch_wet / deposit_ALD kernels (tools/time_wet_ald.py)
TAG=${1:-sqald}
N=${2:-4096}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 tools/time_wet_ald.py $N > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
from collections import OrderedDict
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
disp = OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(x in n for x in ("k_ald_coat", "k_wet_step", "k_ald_seeds")): continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": n.split("(")[0].replace("void cgk::", "").replace("cgk::", "")[:24]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
for d in disp.values():
    w = max(1.0, d.get("SQ_WAVES", 1))
    print("%-24s waves %8d  per wave: valu %7.0f salu %6.0f lds %6.0f vmrd %6.0f  wavecyc %9.0f  wait_inst %3.0f%%" % (
        d["name"], w, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w, d.get("SQ_INSTS_VMEM_RD", 0) / w,
        d.get("SQ_WAVE_CYCLES", 0) / w, 100 * d.get("SQ_WAIT_INST_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1))))
PY
find $OUT -name "*.csv" -size +2M -delete
