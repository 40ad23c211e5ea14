set -u
R=$PWD; OUT=$R/gpurun_out; A=$R/tools/probe/ab
( for l in "" $A/libntmix.so $A/libr04.so; do echo "digest lib=${l:-tree}"; VPD_LIB_PATH=$l python3 tools/step_digest.py 2>/dev/null; done ) > $OUT/r05_digest3.txt 2>&1
python3 -m pytest tests/test_apply_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT/r05_digest3.txt
bash tools/ab_env.sh "new:" "ntmix:VPD_LIB_PATH=$A/libntmix.so" "r04:VPD_LIB_PATH=$A/libr04.so" >> $OUT/r05_digest3.txt 2>&1
cat $OUT/r05_digest3.txt
