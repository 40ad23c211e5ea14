# same-box comparison of this tree's library with round 3's final one (tools/probe/ab/libr03.so = commit 0cf77b0), alternating runs
set -u
R=$PWD; OUT=$R/gpurun_out; T=$OUT/r04_vs_r03_same_box.txt
: > $T
echo "== train step, 256 crops (bench.py --repeats 3)" >> $T
bash tools/ab_env.sh "r04:" "r03:VPD_LIB_PATH=$R/tools/probe/ab/libr03.so" >> $T 2>&1
echo "== train step, 512 crops" >> $T
AB_EXTRA="--batch 512" bash tools/ab_env.sh "r04:" "r03:VPD_LIB_PATH=$R/tools/probe/ab/libr03.so" >> $T 2>&1
echo "== apply (tools/bench_apply.py --batches 30): forward_resident / loop_host_u8 crops/s" >> $T
for rep in 1 2; do for cfg in "r04:" "r03:VPD_LIB_PATH=$R/tools/probe/ab/libr03.so"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  env $envs python3 tools/bench_apply.py --batches 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label rep$rep:', ' '.join('%s=%.0f' % (k, d[k]) for k in ('forward_resident','loop_resident','loop_host_u8','loop_host_fp32')))
" >> $T 2>&1
done; done
cat $T
