#!/bin/bash
# Kernel traces of several library variants on ONE box: tools/ab_multi.sh <tag> <rounds> ab/lib_x.so ab/lib_y.so ...
# prints clouds/s and the gemm_ws / da_dw / smallm kernel averages per variant and round
set -e -o pipefail
TAG=$1; R=$2; shift 2
LIB=3d-pointcloud-orientation-estimation_amd/pnpp_hip/libpnpp_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for L in "$@"; do
    n=$(basename $L .so)
    cp $L $LIB
    bash tools/quick_trace.sh ${TAG}_${n}_$r > gpurun_out/${TAG}_${n}_$r.log 2>&1 || { echo "== $n r$r FAILED"; tail -5 gpurun_out/${TAG}_${n}_$r.log; continue; }
    echo "== $n r$r $(tail -1 gpurun_out/${TAG}_${n}_$r.log)"
    python3 tools/ab_show.py gpurun_out/${TAG}_${n}_$r/kernel_stats_all_by_shape.csv ${AB_FILTER:-dW} || true
  done
done
cp /tmp/lib_keep.so $LIB
