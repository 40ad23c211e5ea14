set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/driver_bench.json 2> $OUT/driver_bench.err; tail -4 $OUT/driver_bench.err
python3 - <<PY
import json
d=json.loads(open('$OUT/driver_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['repeats']['crops_per_s'], d['roofline']['frac'], d['roofline']['kernel'], d['apply']['loop_vs_graph'], d['cpu_baseline']['value'])
PY
