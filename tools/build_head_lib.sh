#!/bin/bash
# Same-box A/B of the working tree against the last commit: builds HEAD's vpd_amd/csrc in /tmp and copies the library to
# tools/probe/ab/libold.so (in-tree, so gpurun ships it; git-ignored).  Then: tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$PWD/tools/probe/ab/libold.so"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/vpd_head && mkdir -p /tmp/vpd_head
git -C "$R" archive ${1:-HEAD} vpd_amd/csrc include | tar -x -C /tmp/vpd_head
make -C /tmp/vpd_head/vpd_amd/csrc -j4 > /tmp/vpd_head/build.log 2>&1 || { tail -20 /tmp/vpd_head/build.log; exit 1; }
mkdir -p "$R/tools/probe/ab" && cp /tmp/vpd_head/vpd_amd/libvpdhip.so "$R/tools/probe/ab/libold.so"
ls -la "$R/tools/probe/ab/libold.so"
