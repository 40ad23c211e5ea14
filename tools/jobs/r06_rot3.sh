set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_pws_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu > $OUT/r06_rot_tests.log 2>&1; echo "tests rc $?"; tail -6 $OUT/r06_rot_tests.log
( echo "digest rotation on (default):"; python3 tools/step_digest.py 2>/dev/null; echo "digest rotation on, generic K loop:"; VPD_PWS_GEO=0 python3 tools/step_digest.py 2>/dev/null; echo "digest rotation off:"; VPD_PWS_ROT=0 python3 tools/step_digest.py 2>/dev/null ) > $OUT/r06_ab_chunk_rotation.txt 2>&1
bash tools/ab_env.sh "rot_on:" "rot_off:VPD_PWS_ROT=0" >> $OUT/r06_ab_chunk_rotation.txt 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "rot_on_512:" "rot_off_512:VPD_PWS_ROT=0" >> $OUT/r06_ab_chunk_rotation.txt 2>&1
cut -c1-200 $OUT/r06_ab_chunk_rotation.txt
