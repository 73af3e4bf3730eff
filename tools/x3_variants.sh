#!/bin/bash
# A/B of library variants on ONE box with split products on: tools/x3_variants.sh <variant>...  (abx/lib_<variant>.so; abx/lib_stamps.so for the stamps)
set -o pipefail
LIB=3d-pointcloud-orientation-estimation_amd/pnpp_hip/libpnpp_hip.so
mkdir -p gpurun_out/x3
cp $LIB /tmp/lib_keep.so
export PNPP_SPLIT_PRODUCTS=1
for v in "$@"; do
  cp abx/lib_$v.so $LIB
  PNPP_BENCH_DUMP=gpurun_out/x3/tab_$v.txt timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/x3/b_$v.json 2> gpurun_out/x3/b_$v.err || { echo "== $v FAILED"; tail -3 gpurun_out/x3/b_$v.err; continue; }
  echo "== $v $(python3 -c "import json;d=json.loads(open('gpurun_out/x3/b_$v.json').readline());print(round(d['value']), d['ms_per_step'])")"
  grep -E "wsd3|wsp3|wsf3|wsx3" gpurun_out/x3/tab_$v.txt
done
if [ -f abx/lib_stamps.so ]; then
  cp abx/lib_stamps.so $LIB
  timeout -k 10 200 python3 tools/wsd3_stamps.py > gpurun_out/x3/stamps.txt 2>&1; cat gpurun_out/x3/stamps.txt
fi
cp /tmp/lib_keep.so $LIB
