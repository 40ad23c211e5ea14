# final tree vs round 3's library on one box + the 1,024-crop bench line
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/jobs/r04_vs_r03.sh | tail -30
python3 bench.py --batch 1024 --no-cpu-baseline --no-apply --repeats 3 --steps 50 --warmup 10 > $OUT/r04_bench_1024.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/r04_bench_1024.json').read().strip().splitlines()[-1])
print('1024: %.1f crops/s %.3f ms step_frac %.3f matrix_frac %.3f dom %s %.0f TF' % (d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'], d['roofline']['matrix_kernels_frac'], d['roofline']['kernel'], d['roofline']['achieved']))
"
