# transforming loaders (conv_xf.hip): op-level parity first; only then the model tests, the bit-equality of whole steps and the A/B
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "loaders" > $OUT/xf1_ops.log 2>&1 || { tail -30 $OUT/xf1_ops.log; exit 1; }
tail -3 $OUT/xf1_ops.log
for v in 1 0 1; do VPD_CONV_XF=$v timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s/^/XF=$v /"; done | tee $OUT/xf1_digest.txt
[ -n "${XF_SKIP_MODEL:-}" ] || timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_pws_gpu.py -x -q > $OUT/xf1_model.log 2>&1; tail -3 $OUT/xf1_model.log
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" > $OUT/xf1_ab.txt 2>&1
cat $OUT/xf1_ab.txt
