# Round-5 evidence set from ONE box: PMC traffic + MFMA duty, default bench line, kernel trace + summary + gaps (tools/collect_profiles.sh),
# then the other workloads' bench lines and the 1 M-crop apply job.
set -u
R=$PWD; OUT=$R/gpurun_out
bash tools/collect_profiles.sh r05 > $OUT/r05_collect.log 2>&1
tail -4 $OUT/r05_collect.log
python3 bench.py --batch 512 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r05_bench_512.json 2>/dev/null
python3 bench.py --arch resnet18 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r05_bench_resnet18.json 2>/dev/null
python3 bench.py --arch resnet50 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r05_bench_resnet50.json 2>/dev/null
python3 bench.py --config c3 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r05_bench_c3.json 2>/dev/null
python3 bench.py --config c4 --no-cpu-baseline --no-apply --repeats 3 > $OUT/r05_bench_c4.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 > $OUT/r05_apply_bench.json 2>/dev/null
python3 tools/bench_apply.py --batches 30 --crops 1000000 --out_dir /tmp/vpd_apply_out > $OUT/r05_apply_bench_1M.json 2>/dev/null
for f in r05_bench_default r05_bench_512 r05_bench_resnet18 r05_bench_resnet50 r05_bench_c3 r05_bench_c4; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f: %.1f crops/s %.3f ms step_frac %.3f matrix_frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'], d['roofline']['matrix_kernels_frac'] or 0))
"; done
python3 -c "
import json
d=json.loads(open('$OUT/r05_apply_bench_1M.json').read().strip().splitlines()[-1])
print('apply: fwd %.0f loop_u8 %.0f full %s' % (d['forward_resident'], d['loop_host_u8'], {k: d['full_run'][k] for k in ('crops_per_s','videos','seconds') if k in d['full_run']}))
"
# data parallel's price on one GPU: layer3 / layer4 weight gradients in two launches (what VPD_TRAIN_EARLY_BUCKET0 selects) against the merged one
bash tools/ab_env.sh "merged:" "unmerged:VPD_WG_MERGE=0" > $OUT/r05_ab_wg_unmerge.txt 2>&1
# the tree against round 4's library, same box, with whole-step digests
( echo "digest tree:"; python3 tools/step_digest.py 2>/dev/null; echo "digest r04:"; VPD_LIB_PATH=$R/tools/probe/ab/libr04.so python3 tools/step_digest.py 2>/dev/null ) > $OUT/r05_vs_r04_same_box.txt 2>&1
bash tools/ab_env.sh "r05:" "r04:VPD_LIB_PATH=$R/tools/probe/ab/libr04.so" >> $OUT/r05_vs_r04_same_box.txt 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "r05_512:" "r04_512:VPD_LIB_PATH=$R/tools/probe/ab/libr04.so" >> $OUT/r05_vs_r04_same_box.txt 2>&1
# the ResNet-50 step: kernel summary + timeline, and its round-5 paths against the launches they replace
bash tools/jobs/r05_trace50.sh r05_resnet50 > /dev/null 2>&1
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "default:" "no_stream_no_recompute:VPD_CONV1X1_STREAM=0,VPD_BNECK_RECOMPUTE=0" "no_recompute:VPD_BNECK_RECOMPUTE=0" 2>&1 | cut -c1-330 > $OUT/r05_ab_r50_paths.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/r05_gputests.log 2>&1; tail -2 $OUT/r05_gputests.log
