# kernel trace of the default bench and of the ResNet-50 bench -> r05_{step_timeline,launch_gaps}.txt, r05_resnet50_{summary,timeline}.txt
set -u
R=$PWD; OUT=$R/gpurun_out
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rt_prof -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
python3 $R/tools/launch_gaps.py $OUT/rt_prof/p_kernel_trace.csv > $OUT/r05_launch_gaps.txt 2>&1
python3 $R/tools/step_timeline.py $OUT/rt_prof/p_kernel_trace.csv > $OUT/r05_step_timeline.txt 2>&1
rm -rf $OUT/rt_prof
cd $R && bash tools/jobs/r05_trace50.sh r05_resnet50 > /dev/null 2>&1
head -6 $OUT/r05_launch_gaps.txt; tail -1 $OUT/r05_step_timeline.txt; tail -1 $OUT/r05_resnet50_timeline.txt
