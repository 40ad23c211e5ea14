# kernel trace of the ResNet-50 bench -> per-kernel summary + timeline: r05_trace50.sh <tag> [ENV=VAL ...]
set -u
R=$PWD; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python3 $R/bench.py --arch resnet50 --steps 10 --warmup 3 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply --no-parity > $OUT/${TAG}_bench.json 2> /dev/null
python3 $R/tools/step_timeline.py $OUT/${TAG}_prof/p_kernel_trace.csv > $OUT/${TAG}_timeline.txt 2>&1
python3 $R/tools/prof_summary.py $OUT/${TAG}_prof/p_kernel_stats.csv 13 60 > $OUT/${TAG}_summary.txt
rm -rf $OUT/${TAG}_prof
head -50 $OUT/${TAG}_summary.txt
