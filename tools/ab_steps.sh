#!/bin/bash
# A/B of per-step times: the flow's steps one by one (tools/step_times.py), with and without an environment switch.
# usage: tools/ab_steps.sh "CG_NO_LEAN=1" [n] [flow]
SW=${1:-CG_NO_LEAN=1}; N=${2:-16384}; FLOW=${3:-c4}
mkdir -p gpurun_out
for rep in 1 2; do
  echo "== default";  NOSTATS=1 COMPACT=1 REPS=3 timeout -k 10 200 python3 tools/step_times.py $N $FLOW | tail -2 || exit 1
  echo "== $SW";      env $SW NOSTATS=1 COMPACT=1 REPS=3 timeout -k 10 200 python3 tools/step_times.py $N $FLOW | tail -2 || exit 1
done
