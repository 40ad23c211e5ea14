#!/bin/bash
# Same-box A/B of environment switches: tools/ab_env.sh "<label>:<ENV=..,ENV=..>" ...   ("<label>:" alone = defaults)
# Each configuration runs bench.py twice, alternating, and prints crops/s + the per-class table of each run.
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" 2>/dev/null
ARGS="--no-cpu-baseline --no-apply --no-parity --repeats 3 --steps 100 --warmup 20 ${AB_EXTRA:-}"      # AB_EXTRA: e.g. "--arch resnet50" or "--batch 512"
for rep in 1 2; do
  for cfg in "$@"; do
    label=${cfg%%:*}; envs=${cfg#*:}
    env $(echo "$envs" | tr ',' ' ') python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('$label rep$rep: %.1f crops/s %.3f ms | ' % (d['value'], d['ms_per_step']) + ' '.join('%s=%.0fus/%.0fTF' % (n.split('_kernel')[0][-12:], v['ms_per_step']*1e3, v['tflops']) for n,v in k.items()))
"
  done
done
