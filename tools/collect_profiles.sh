#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_{kernel_stats.csv,summary.txt,pmc_traffic.json,mfma_util.json,bench_*.json}
# Kernel trace and counters are SEPARATE runs (gpurun refuses --pmc together with trace domains other than kernel-trace;
# FETCH_SIZE and WRITE_SIZE do not fit one pass).  The profiled command is the same bench.py invocation every time.
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out
ARGS="--steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply --no-parity"
cd /tmp && export TMPDIR=/tmp
export VPD_PROFILE_TAG=$TAG
# 1. counters first: the default bench run below then reads the traffic figure collected on THIS box (roofline.traffic_source)
PARGS="--steps 4 --warmup 2 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply --no-parity"
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o pmc --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/${TAG}_pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2>&1
python3 $R/tools/pmc_mfma_util.py $OUT/${TAG}_pmc_mfma $OUT/${TAG}_mfma_util.json > $OUT/${TAG}_mfma_util.txt
# 2. the default bench line (with cpu_baseline and the apply block) and the kernel trace of the same command
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python3 $R/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $OUT/${TAG}_prof/p_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 $R/tools/launch_gaps.py $OUT/${TAG}_prof/p_kernel_trace.csv > $OUT/${TAG}_launch_gaps.txt 2>&1
python3 $R/tools/step_timeline.py $OUT/${TAG}_prof/p_kernel_trace.csv > $OUT/${TAG}_step_timeline.txt 2>&1
python3 $R/tools/prof_summary.py $OUT/${TAG}_kernel_stats.csv 25 60 > $OUT/${TAG}_summary.txt
python3 $R/tools/bandwidth_table.py $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_bandwidth.txt 2>/dev/null
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_prof
tail -3 $OUT/${TAG}_summary.txt; head -5 $OUT/${TAG}_mfma_util.txt
