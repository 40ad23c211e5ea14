set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export VPD_CONV_XF=$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT/xft_$v -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
  python3 $R/tools/step_timeline.py $OUT/xft_$v/p_kernel_trace.csv > $OUT/xf_timeline_$v.txt
  rm -rf $OUT/xft_$v
done
