set -u
R=$PWD; OUT=$R/gpurun_out
python3 bench.py --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04a_bench.json 2> $OUT/r04a_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04a_prof -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
cd $R
ls $OUT/r04a_prof
python3 tools/launch_gaps.py $OUT/r04a_prof/p_kernel_trace.csv > $OUT/r04a_gaps.txt
python3 tools/step_timeline.py $OUT/r04a_prof/p_kernel_trace.csv > $OUT/r04a_timeline.txt
python3 tools/prof_summary.py $OUT/r04a_prof/p_kernel_stats.csv 25 60 > $OUT/r04a_summary.txt
rm -rf $OUT/r04a_prof

cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/r04a_applyprof -o p -- python3 $R/tools/apply_loop_probe.py 12 > $OUT/r04a_apply_probe.txt 2>&1
cd $R
python3 tools/apply_loop_probe.py --analyse $OUT/r04a_applyprof/p >> $OUT/r04a_apply_probe.txt 2>&1
head -3 $OUT/r04a_applyprof/p_memory_copy_trace.csv >> $OUT/r04a_apply_probe.txt
rm -rf $OUT/r04a_applyprof
python3 tools/apply_loop_probe.py 12 >> $OUT/r04a_apply_probe.txt 2>&1
tail -25 $OUT/r04a_apply_probe.txt
cat $OUT/r04a_gaps.txt; head -c 600 $OUT/r04a_bench.json
