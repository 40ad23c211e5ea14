set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2; do timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s#^#tree run$i #"; done | tee $OUT/bnreg_digest.txt
VPD_CONV_XF=0 timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s#^#tree XF=0 #" | tee -a $OUT/bnreg_digest.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/bnreg_tests.log 2>&1; tail -2 $OUT/bnreg_tests.log
bash tools/ab_env.sh "new:" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" > $OUT/bnreg_ab.txt 2>&1; cut -c1-100 $OUT/bnreg_ab.txt
