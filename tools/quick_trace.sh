#!/bin/bash
# kernel trace of the replayed step, all kernels by shape: tools/quick_trace.sh <tag> [extra bench args]
set -e -o pipefail
TAG=${1:-qt}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-bf16-variant --no-mfma-variant "$@" > $OUT/trace.log 2>&1
python3 tools/summarize_rocprof.py stats $OUT/trace $OUT/kernel_stats.csv
rm -rf $OUT/trace
grep -E '^\{' $OUT/trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('clouds/s', round(d['value']), 'ms', round(d['ms_per_step'],4))"
