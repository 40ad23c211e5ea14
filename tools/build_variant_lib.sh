#!/bin/bash
# Build the WORKING TREE's library with extra compiler flags into tools/probe/ab/lib<name>.so (in-tree: gpurun ships it; git-ignored):
#   tools/build_variant_lib.sh stamps -DPWS_STAMPS        then   VPD_LIB_PATH=$PWD/tools/probe/ab/libstamps.so python3 bench.py ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/vpd_variant_$NAME
rm -rf $D && mkdir -p $D/vpd_amd && cp -r $R/vpd_amd/csrc $D/vpd_amd/csrc && cp -r $R/include $D/include && rm -rf $D/vpd_amd/csrc/build
make -C $D/vpd_amd/csrc -j6 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $*" > $D/build.log 2>&1 || { grep -E "error" -A5 $D/build.log | head -30; exit 1; }
mkdir -p "$R/tools/probe/ab" && cp $D/vpd_amd/libvpdhip.so "$R/tools/probe/ab/lib$NAME.so"
ls -la "$R/tools/probe/ab/lib$NAME.so"
