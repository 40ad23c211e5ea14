set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for lib in "" "$R/tools/probe/ab/libold.so"; do VPD_LIB_PATH=$lib timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s#^#lib=${lib:-tree} #"; done | tee $OUT/folds_digest.txt
VPD_ADAM_STEM=0 VPD_POOLBWD_FOLD=0 timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s/^/tree, folds off /" | tee -a $OUT/folds_digest.txt
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_ddp_gpu.py -x -q > $OUT/folds_tests.log 2>&1; tail -3 $OUT/folds_tests.log
bash tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" > $OUT/folds_ab.txt 2>&1
cut -c1-120 $OUT/folds_ab.txt
AB_EXTRA="--config c3" bash tools/ab_env.sh "new_c3:" "old_c3:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" 2>&1 | cut -c1-120 | tee $OUT/folds_ab_c3.txt
