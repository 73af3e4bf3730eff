#!/bin/bash
# Build a variant of libpnpp_hip.so with extra -D flags on gemm_wsp_kernels.hip: tools/build_wsp_variant.sh <name> [-DFLAG=..]...
# -> ab/lib_<name>.so (ab/ is git-ignored; it travels to the GPU box with the snapshot)
set -e -o pipefail
NAME=$1; shift
PKG=3d-pointcloud-orientation-estimation_amd
mkdir -p ab/obj_$NAME
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -Wno-pass-failed -Iinclude "$@" \
    -c $PKG/csrc/gemm_wsp_kernels.hip -o ab/obj_$NAME/gemm_wsp_kernels.o
OBJS=$(ls $PKG/csrc/_obj/*.o | grep -v "/gemm_wsp_kernels.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$NAME.so ab/obj_$NAME/gemm_wsp_kernels.o $OBJS
echo built ab/lib_$NAME.so
