# Round-6 evidence set from ONE box: PMC traffic + MFMA duty, default bench line (parity, cpu_baseline, apply), kernel trace + summary + gaps
# (tools/collect_profiles.sh), the other workloads' bench lines, fp16, the 1 M-crop apply job, same-box A/B against round 5's library, GPU suite.
set -u
R=$PWD; OUT=$R/gpurun_out
bash tools/collect_profiles.sh r06 > $OUT/r06_collect.log 2>&1
tail -4 $OUT/r06_collect.log
B="--no-cpu-baseline --no-apply --no-parity --repeats 3"
python3 bench.py --batch 512 $B > $OUT/r06_bench_512.json 2>/dev/null
python3 bench.py --batch 1024 $B > $OUT/r06_bench_1024.json 2>/dev/null
python3 bench.py --arch resnet18 $B > $OUT/r06_bench_resnet18.json 2>/dev/null
python3 bench.py --arch resnet50 $B > $OUT/r06_bench_resnet50.json 2>/dev/null
python3 bench.py --config c3 $B > $OUT/r06_bench_c3.json 2>/dev/null
python3 bench.py --config c4 $B > $OUT/r06_bench_c4.json 2>/dev/null
python3 bench.py --dtype fp16 $B > $OUT/r06_bench_fp16.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 > $OUT/r06_apply_bench.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 --dtype fp16 > $OUT/r06_apply_bench_fp16.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 --crops 1000000 --out_dir /tmp/vpd_apply_out > $OUT/r06_apply_bench_1M.json 2>/dev/null
for f in r06_bench_default r06_bench_512 r06_bench_1024 r06_bench_resnet18 r06_bench_resnet50 r06_bench_c3 r06_bench_c4 r06_bench_fp16; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f: %.1f crops/s %.3f ms step_frac %.3f matrix_frac %.3f dom %.3f' % (d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'], d['roofline']['matrix_kernels_frac'] or 0, d['roofline']['frac']), d.get('parity'))
"; done
# the tree against round 5's library, same box, whole-step digests first (they differ: chunk rotation changes the fp32 summation order)
export VPD_LIB_ALLOW_ABI=1
( echo "digest tree:"; python3 tools/step_digest.py 2>/dev/null; echo "digest r05:"; VPD_LIB_PATH=$R/tools/probe/ab/libr05.so python3 tools/step_digest.py 2>/dev/null ) > $OUT/r06_vs_r05_same_box.txt 2>&1
bash tools/ab_env.sh "r06:" "r05:VPD_LIB_PATH=$R/tools/probe/ab/libr05.so" >> $OUT/r06_vs_r05_same_box.txt 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "r06_512:" "r05_512:VPD_LIB_PATH=$R/tools/probe/ab/libr05.so" >> $OUT/r06_vs_r05_same_box.txt 2>&1
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "r06_r50:" "r05_r50:VPD_LIB_PATH=$R/tools/probe/ab/libr05.so" >> $OUT/r06_vs_r05_same_box.txt 2>&1
unset VPD_LIB_ALLOW_ABI
cut -c1-120 $OUT/r06_vs_r05_same_box.txt
bash tools/jobs/r05_trace50.sh r06_resnet50 > /dev/null 2>&1
