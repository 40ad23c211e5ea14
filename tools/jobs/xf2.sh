set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "loaders" > $OUT/xf2_ops.log 2>&1 || { tail -30 $OUT/xf2_ops.log; exit 1; }
tail -2 $OUT/xf2_ops.log
for v in 1 0; do VPD_CONV_XF=$v timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s/^/XF=$v /"; done | tee $OUT/xf2_digest.txt
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" ${XF_EXTRA_CFGS:-} > $OUT/xf2_ab.txt 2>&1
cut -c1-200 $OUT/xf2_ab.txt
bash tools/jobs/xf_stamps.sh | grep -A 15 "xf stamps"
