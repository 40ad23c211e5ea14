# usage: bash tools/jobs/ab_generic.sh <tag> [pytest files...]   -- tests, then same-box A/B of the tree's library against tools/probe/ab/libold.so
set -u
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out
if [ $# -gt 0 ]; then python -m pytest "$@" -x -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log; fi
bash tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
