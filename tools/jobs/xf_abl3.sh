set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" "no_a1:VPD_XF_ABLATE=128" "no_mask:VPD_XF_ABLATE=256" "neither:VPD_XF_ABLATE=384" > $OUT/xf_abl3.txt 2>&1
cut -c1-150 $OUT/xf_abl3.txt
