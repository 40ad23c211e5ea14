# the launch-floor probe (events + in-kernel stamps, null and created stream), the same binary under
# rocprofv3 --kernel-trace, and the train step on the null stream against a created stream.
set -u
R=$PWD; OUT=$R/gpurun_out
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r05_floor_probe.txt 2>&1
timeout -k 10 240 $R/tools/probe/floor_probe >> $OUT/r05_floor_probe.txt 2>&1 && \
timeout -k 10 240 $R/tools/probe/floor_probe --created >> $OUT/r05_floor_probe.txt 2>&1 && \
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r05_floor_prof -o p -- $R/tools/probe/floor_probe --noevents --reps 7 > $OUT/r05_floor_probe_rocprof_stdout.txt 2>&1 ) && \
python3 tools/probe/floor_table.py $(find $OUT/r05_floor_prof -name 'p_kernel_trace.csv' | head -1) 7 > $OUT/r05_floor_probe_rocprof.txt 2>&1
tail -5 $OUT/r05_floor_probe_rocprof.txt
rm -rf $OUT/r05_floor_prof
bash tools/ab_env.sh "null:" "created:VPD_BENCH_STREAM=created" > $OUT/r05_stream_ab.txt 2>&1
cat $OUT/r05_stream_ab.txt
