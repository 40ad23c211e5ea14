set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" "nostore:VPD_XF_ABLATE=1" "nomath:VPD_XF_ABLATE=2" "nofin:VPD_XF_ABLATE=4" "none:VPD_XF_ABLATE=7" > $OUT/xf_ablate.txt 2>&1
cat $OUT/xf_ablate.txt
