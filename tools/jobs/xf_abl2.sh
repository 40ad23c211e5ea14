set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" "one:VPD_XF_ABLATE=32" "noput:VPD_XF_ABLATE=64" "noput_nostore:VPD_XF_ABLATE=65" > $OUT/xf_abl2.txt 2>&1
cut -c1-150 $OUT/xf_abl2.txt
