#!/bin/bash
# Runs the GPU test files one process at a time, in dependency order; stops at the first timeout.
# usage: tests/run_gpu.sh <logfile> [pytest args...]
log=$1; shift
: > "$log"
for f in tests/test_gpu_index.py tests/test_gpu_head_loss.py tests/test_gpu_sa.py tests/test_gpu_e2e.py tests/test_gpu_f3.py tests/test_gpu_pt.py tests/test_gpu_train.py; do
  echo "=== $f" >> "$log"
  timeout -k 10 300 python -m pytest "$f" -m gpu -q -p no:cacheprovider --maxfail=6 -s "$@" >> "$log" 2>&1
  rc=$?
  echo "=== rc $rc" >> "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $f, stopping" >> "$log"; exit 1; fi
done
grep -E "^(=== |FAILED|ERROR|[0-9]+ (passed|failed))|passed|failed" "$log" | tail -40
