#!/bin/bash
# rocprofv3 kernel-trace of the train step under two environments, per-kernel averages side by side:
#   tools/prof_ab.sh <tag> "A:ENV=..,ENV=.." "B:ENV=.."     -> gpurun_out/<tag>_{A,B}_summary.txt
set -u
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out
ARGS="--steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply --no-parity"
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${label}_prof -o p -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  for kv in $(echo "$envs" | tr ',' ' '); do unset "${kv%%=*}"; done
  cp $OUT/${TAG}_${label}_prof/p_kernel_stats.csv $OUT/${TAG}_${label}_kernel_stats.csv
  python3 $R/tools/prof_summary.py $OUT/${TAG}_${label}_kernel_stats.csv 25 70 > $OUT/${TAG}_${label}_summary.txt
  rm -rf $OUT/${TAG}_${label}_prof
done
