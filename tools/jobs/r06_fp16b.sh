set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 1000 python -m pytest tests/test_fp16_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "fp16 or loss_scaler" > $OUT/r06_fp16_tests.log 2>&1; echo "tests rc $?"; tail -40 $OUT/r06_fp16_tests.log | cut -c1-400
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("OUT","gpurun_out")+"/parity_fp16_*.json")):
    d=json.load(open(f))
    print(os.path.basename(f), {k:(round(v,5) if isinstance(v,float) else v) for k,v in d.items() if k in ("emb_eval_per_sample_max","loss_train","grad_flat_err","grad_err_max","grad_cos_min","running_stats_rel_l2_max","epoch_traj","hip_vs_reference","grad_norm_ratio_by_stage","emb_train_per_sample_max")})
PY
