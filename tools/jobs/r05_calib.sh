# In-situ calibration of the launch floor: n extra empty 256-block launches behind each of the 32 BatchNorm-forward launches of the step.
set -u
R=$PWD; OUT=$R/gpurun_out; L=$R/tools/probe/ab/libcalib.so
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r05_calib_ab.txt 2>&1
bash tools/ab_env.sh "n0:VPD_LIB_PATH=$L" "n1:VPD_LIB_PATH=$L,VPD_CALIB_EMPTY=1" "n2:VPD_LIB_PATH=$L,VPD_CALIB_EMPTY=2" "n4:VPD_LIB_PATH=$L,VPD_CALIB_EMPTY=4" >> $OUT/r05_calib_ab.txt 2>&1
cat $OUT/r05_calib_ab.txt
cd /tmp && export TMPDIR=/tmp
VPD_LIB_PATH=$L VPD_CALIB_EMPTY=2 rocprofv3 --kernel-trace --output-format csv -d $OUT/r05_calib_prof -o p -- python3 $R/bench.py --steps 12 --warmup 3 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
python3 $R/tools/step_timeline.py $(find $OUT/r05_calib_prof -name p_kernel_trace.csv | head -1) > $OUT/r05_calib_timeline.txt 2>&1
rm -rf $OUT/r05_calib_prof
grep -c calib $OUT/r05_calib_timeline.txt
