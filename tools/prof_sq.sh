#!/bin/bash
# per-dispatch SQ counters of one flow pass
TAG=${1:-sq}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc -- python3 tools/step_times.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
from collections import OrderedDict
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "cgk::" not in n or any(x in n for x in ("mask", "checksum", "gather", "scatter", "max_top", "halo")): continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": n.split("(")[0].replace("void cgk::", "")[:40]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = list(disp.values()); half = len(ds) // 2
for k, d in enumerate(ds[half:]):
    print("%3d %-40s waves %8d valu %6.1fM vmrd %5.2fM vmwr %5.2fM wavecyc %7.2fG wait_any %4.0f%% wait_inst %4.0f%% busy %6.1fM" % (
        k, d["name"], d.get("SQ_WAVES", 0), d.get("SQ_INSTS_VALU", 0) / 1e6, d.get("SQ_INSTS_VMEM_RD", 0) / 1e6, d.get("SQ_INSTS_VMEM_WR", 0) / 1e6,
        d.get("SQ_WAVE_CYCLES", 0) / 1e9, 100 * d.get("SQ_WAIT_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)),
        100 * d.get("SQ_WAIT_INST_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), d.get("SQ_BUSY_CYCLES", 0) / 1e6))
PY
find $OUT -name "*.csv" -size +2M -delete
