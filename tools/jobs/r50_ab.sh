set -u
R=$PWD; OUT=$R/gpurun_out
python -m pytest tests -m gpu -x -q > $OUT/r04j_gputests.log 2>&1; tail -3 $OUT/r04j_gputests.log
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "sums_off:" "sums_on:VPD_DGRAD_SUMS_BNECK=1" > $OUT/r04j_r50_ab.txt 2>&1
cat $OUT/r04j_r50_ab.txt
