set -u
R=$PWD; OUT=$R/gpurun_out
bash tools/prof_ab.sh r06_rot "rot0:VPD_PWS_ROT=0" "rot1:VPD_PWS_ROT=1" > /dev/null 2>&1
cd $R
paste <(cut -c1-90 $OUT/r06_rot_rot0_summary.txt) <(cut -c66-90 $OUT/r06_rot_rot1_summary.txt) | head -45
bash tools/ab_env.sh "rot0:" "rot1:VPD_PWS_ROT=1" 2>&1 | cut -c1-60
VPD_PWS_ROT=1 timeout -k 10 900 python -m pytest tests/test_pws_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu > $OUT/r06_rot_tests.log 2>&1; tail -15 $OUT/r06_rot_tests.log
