#!/bin/bash
# ms_per_step of several short bench runs on one box (driver-style: 20 steps after 5 warm-up steps) beside a long one:
#   tools/bench_repeat.sh <tag> [extra bench args]     -> gpurun_out/<tag>.txt
TAG=$1; shift
OUT=gpurun_out/$TAG.txt
: > $OUT
for i in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('K=20  W=5  ms_per_step %.4f  clouds/s %.0f' % (d['ms_per_step'], d['value']))" >> $OUT
done
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('K=200 W=20 ms_per_step %.4f  clouds/s %.0f' % (d['ms_per_step'], d['value']))" >> $OUT
cat $OUT
