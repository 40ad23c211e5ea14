# Round 6: compile-time-geometry K loop of conv3x3_pws_kernel (VPD_PWS_GEO=1, default) against the generic one (=0): operator tests,
# whole-step digests (must be equal: every accumulator sees the same K order), alternating bench runs, 512 crops, apply.
set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_pws_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $OUT/r06_geo_tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/r06_geo_tests.log
( echo "digest geo:"; python3 tools/step_digest.py 2>/dev/null; echo "digest generic:"; VPD_PWS_GEO=0 python3 tools/step_digest.py 2>/dev/null ) > $OUT/r06_ab_geo.txt 2>&1
bash tools/ab_env.sh "geo:" "generic:VPD_PWS_GEO=0" >> $OUT/r06_ab_geo.txt 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "geo_512:" "generic_512:VPD_PWS_GEO=0" >> $OUT/r06_ab_geo.txt 2>&1
cut -c1-400 $OUT/r06_ab_geo.txt
