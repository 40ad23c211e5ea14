set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
AB_EXTRA="--batch 512" bash tools/ab_env.sh "xf512:VPD_CONV_XF=1" "base512:VPD_CONV_XF=0" 2>&1 | cut -c1-120 | tee $OUT/xf_wide_512.txt
AB_EXTRA="--arch resnet18" bash tools/ab_env.sh "xf_r18:VPD_CONV_XF=1" "base_r18:VPD_CONV_XF=0" 2>&1 | cut -c1-120 | tee $OUT/xf_wide_r18.txt
AB_EXTRA="--config c3" bash tools/ab_env.sh "xf_c3:VPD_CONV_XF=1" "base_c3:VPD_CONV_XF=0" 2>&1 | cut -c1-120 | tee $OUT/xf_wide_c3.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/xf_wide_tests.log 2>&1; tail -3 $OUT/xf_wide_tests.log
