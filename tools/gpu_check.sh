#!/bin/bash
# parity tests, then the 16384^2 bench; results under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread ${DESELECT:+--deselect "$DESELECT"} 2>&1 | tail -5 || exit 1
timeout -k 10 400 python -u bench.py ${BENCH_ARGS:---no-cpu} > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "flowGB/s", d["whole_flow_algorithmic_GBps"], "chk", d["checksum"])
for k,v in d["kernels"].items(): print("  %-28s %s" % (k, v))
PY
