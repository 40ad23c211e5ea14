set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
python3 $R/tools/step_timeline.py $OUT/tl/p_kernel_trace.csv > $OUT/r04_step_timeline.txt
python3 $R/tools/launch_gaps.py $OUT/tl/p_kernel_trace.csv > $OUT/r04_launch_gaps_b.txt
rm -rf $OUT/tl; tail -1 $OUT/r04_step_timeline.txt; head -6 $OUT/r04_launch_gaps_b.txt
