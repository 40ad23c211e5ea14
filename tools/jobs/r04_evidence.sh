# Round-4 evidence set from ONE box: PMC traffic + MFMA duty, default bench line, kernel trace + summary + gaps (tools/collect_profiles.sh),
# then the other workloads' bench lines and the 1 M-crop apply job.
set -u
R=$PWD; OUT=$R/gpurun_out
bash tools/collect_profiles.sh r04 > $OUT/r04_collect.log 2>&1
tail -4 $OUT/r04_collect.log
python3 bench.py --batch 512 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04_bench_512.json 2>/dev/null
python3 bench.py --arch resnet18 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04_bench_resnet18.json 2>/dev/null
python3 bench.py --arch resnet50 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04_bench_resnet50.json 2>/dev/null
python3 bench.py --config c3 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04_bench_c3.json 2>/dev/null
python3 bench.py --config c4 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r04_bench_c4.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 > $OUT/r04_apply_bench.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 --crops 1000000 --out_dir /tmp/vpd_apply_out > $OUT/r04_apply_bench_1M.json 2>/dev/null
for f in r04_bench_default r04_bench_512 r04_bench_resnet18 r04_bench_resnet50 r04_bench_c3 r04_bench_c4; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f: %.1f crops/s %.3f ms step_frac %.3f matrix_frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'], d['roofline']['matrix_kernels_frac'] or 0))
"; done
python3 -c "
import json
d=json.loads(open('$OUT/r04_apply_bench_1M.json').read().strip().splitlines()[-1])
print('apply: fwd %.0f loop_u8 %.0f full %s' % (d['forward_resident'], d['loop_host_u8'], {k: d['full_run'][k] for k in ('crops_per_s','videos','seconds') if k in d['full_run']}))
"
