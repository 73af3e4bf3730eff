#!/bin/bash
# Build a variant of libpnpp_hip.so with extra -D flags on ONE source: tools/build_variant_src.sh <name> <source.hip> [-DFLAG=..]...
# -> ab/lib_<name>.so (ab/ is git-ignored; it travels to the GPU box with the snapshot).  Build the library itself first.
set -e -o pipefail
NAME=$1; SRC=$2; shift 2
PKG=3d-pointcloud-orientation-estimation_amd
O=$(basename $SRC .hip).o
mkdir -p ab/obj_$NAME
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -Wno-pass-failed -Iinclude "$@" \
    -c $PKG/csrc/$SRC -o ab/obj_$NAME/$O
OBJS=$(ls $PKG/csrc/_obj/*.o | grep -v "/$O")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$NAME.so ab/obj_$NAME/$O $OBJS
echo built ab/lib_$NAME.so
