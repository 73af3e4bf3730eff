#!/bin/bash
# kernel trace of the step for several values of an environment switch on ONE box: tools/stagger_sweep.sh <tag> <ENVVAR> <filter> v1 v2 ...
TAG=$1; VAR=$2; FLT=$3; shift 3
for v in "$@"; do
  export $VAR=$v
  bash tools/quick_trace.sh ${TAG}_$v > gpurun_out/${TAG}_$v.log 2>&1 || { echo "== $v FAILED"; tail -3 gpurun_out/${TAG}_$v.log; continue; }
  echo "== $VAR=$v $(tail -1 gpurun_out/${TAG}_$v.log)"
  python3 tools/ab_show.py gpurun_out/${TAG}_$v/kernel_stats_all_by_shape.csv $FLT
done
