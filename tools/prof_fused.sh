#!/bin/bash
# SQ counters of the fused vertical launches of one C4 pass (rocprofv3 --pmc, one pass)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/fusedpmc
rm -rf $OUT; mkdir -p $OUT
REPS=1 timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 tools/fused_times.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_vertical_fused" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in rows.items():
    w = c.get("SQ_WAVES", 1)
    print("dispatch %s waves %d per wave: valu %.0f salu %.0f vmrd %.1f vmwr %.1f wavecyc %.0f wait_inst %.0f%%" % (
        d, w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_VMEM_RD", 0) / w, c.get("SQ_INSTS_VMEM_WR", 0) / w,
        c.get("SQ_WAVE_CYCLES", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1))))
PY
find $OUT -name "*.csv" -size +2M -delete
