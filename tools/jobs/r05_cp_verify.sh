# default tree (cache-policy mix) vs round 4's library: whole-step digests (must be identical), GPU suite, train + apply + other batch sizes
set -u
R=$PWD; OUT=$R/gpurun_out; L4=$R/tools/probe/ab/libr04.so
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r05_cp_verify.txt 2>&1
( echo "digest new:"; python3 tools/step_digest.py; echo "digest r04:"; VPD_LIB_PATH=$L4 python3 tools/step_digest.py; echo "digest new resnet50:"; python3 tools/step_digest.py --arch resnet50 --batch 64; echo "digest r04 resnet50:"; VPD_LIB_PATH=$L4 python3 tools/step_digest.py --arch resnet50 --batch 64 ) >> $OUT/r05_cp_verify.txt 2>&1
python3 -m pytest tests -m gpu -x -q > $OUT/r05_gputests_mid.log 2>&1; tail -3 $OUT/r05_gputests_mid.log >> $OUT/r05_cp_verify.txt
bash tools/ab_env.sh "new:" "r04:VPD_LIB_PATH=$L4" >> $OUT/r05_cp_verify.txt 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "new512:" "r04_512:VPD_LIB_PATH=$L4" >> $OUT/r05_cp_verify.txt 2>&1
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "new_r50:" "r04_r50:VPD_LIB_PATH=$L4" >> $OUT/r05_cp_verify.txt 2>&1
for l in "" $L4; do VPD_LIB_PATH=$l python3 tools/bench_apply.py 2>/dev/null | tail -1 >> $OUT/r05_cp_verify.txt; done
cat $OUT/r05_cp_verify.txt
