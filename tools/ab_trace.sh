#!/bin/bash
# A/B kernel traces on ONE box (run-to-run differences between boxes are a few per cent): tools/ab_trace.sh <tag> libA.so libB.so [rounds] ["extra bench args"]
set -e -o pipefail
TAG=$1; A=$2; B=$3; R=${4:-2}; EXTRA=${5:-}
LIB=3d-pointcloud-orientation-estimation_amd/pnpp_hip/libpnpp_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then cp $A $LIB; else cp $B $LIB; fi
    bash tools/quick_trace.sh ${TAG}_${v}${r} $EXTRA > gpurun_out/${TAG}_${v}${r}.log 2>&1
    echo "== $v$r $(tail -1 gpurun_out/${TAG}_${v}${r}.log)"
    grep "gemm_ws" gpurun_out/${TAG}_${v}${r}/kernel_stats.csv | awk -F, '{printf "%s %s | ", $(NF-1), ""} END {print ""}'
  done
done
cp /tmp/lib_keep.so $LIB
