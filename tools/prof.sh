#!/bin/bash
# rocprofv3 passes over bench.py; each pass is its own run (counters are not split over passes).
# usage: tools/prof.sh <tag> [bench args]
TAG=${1:-prof}; shift
ARGS=${@:---no-cpu}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
python3 tools/prof_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
# keep only the small summaries
find $OUT -name "*.csv" -size +2M -delete
