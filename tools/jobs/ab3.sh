# usage: bash tools/jobs/ab3.sh <tag> "<label:ENV=..>" ... -- [pytest files]
set -u
TAG=$1; shift
CFGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CFGS+=("$1"); shift; done
[ $# -gt 0 ] && shift
R=$PWD; OUT=$R/gpurun_out
if [ $# -gt 0 ]; then python -m pytest "$@" -x -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log; fi
bash tools/ab_env.sh "${CFGS[@]}" > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
