#!/bin/bash
# build the library, then run a command on the GPU box: tools/gr.sh <timeout-s> '<command>'
set -e -o pipefail
T=$1; shift
(cd 3d-pointcloud-orientation-estimation_amd && python -m pnpp_hip.build > /dev/null)
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
