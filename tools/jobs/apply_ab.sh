set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_apply_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do for cfg in "new:" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  env $envs python3 tools/bench_apply.py --batches 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label rep$rep:', ' '.join('%s=%.0f' % (k, d[k]) for k in ('forward_resident','loop_resident','loop_host_u8')))
"
done; done | tee $OUT/apply_ab.txt
