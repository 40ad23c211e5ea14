set -u
R=$PWD; OUT=$R/gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_pws_gpu.py -x -q > $OUT/r04c_ops.log 2>&1; tail -3 $OUT/r04c_ops.log
python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q > $OUT/r04c_model.log 2>&1; tail -3 $OUT/r04c_model.log
bash tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" > $OUT/r04c_ab.txt 2>&1
cat $OUT/r04c_ab.txt
