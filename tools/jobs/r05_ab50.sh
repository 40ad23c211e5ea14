# usage: r05_ab50.sh <tag> <variant lib name> [pytest args...]: ResNet-34 digests of tree + variant (must agree when the change claims so),
# optional tests, same-box A/B of the ResNet-50 step
set -u
R=$PWD; OUT=$R/gpurun_out; TAG=$1; V=$R/tools/probe/ab/lib$2.so; shift; shift
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/$TAG.txt 2>&1
( echo "digest tree:"; python3 tools/step_digest.py 2>/dev/null; echo "digest variant:"; VPD_LIB_PATH=$V python3 tools/step_digest.py 2>/dev/null ) >> $OUT/$TAG.txt 2>&1
if [ $# -gt 0 ]; then timeout -k 10 900 python3 -m pytest "$@" -x -q 2>&1 | tail -3 >> $OUT/$TAG.txt; fi
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$V" 2>&1 | cut -c1-330 >> $OUT/$TAG.txt
cat $OUT/$TAG.txt
