# three fragment sets for the 32-pixel wave tiles (layer4's geo K loop; layer1's conv3x3_c64_persistent_kernel) against two (libsets2.so):
# tests, digests (must be equal: same K order per accumulator), layer4 stamps, alternating runs.
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_three_sets.txt
timeout -k 10 900 python -m pytest tests/test_pws_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $OUT/r06_sets3_tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/r06_sets3_tests.log
( echo "digest three sets (tree):"; python3 tools/step_digest.py 2>/dev/null; echo "digest two sets:"; VPD_LIB_PATH=$R/tools/probe/ab/libsets2.so python3 tools/step_digest.py 2>/dev/null ) > $F 2>&1
for v in stamps stamps2; do echo "=== l4 stamps, lib$v (stamps2 = two sets)" >> $F; BENCH_PWS_LAYERS=l4 VPD_LIB_PATH=$R/tools/probe/ab/lib$v.so python3 tools/bench_pws.py 256 $v 2>&1 | grep -v amdgpu.ids | grep "READY\|K loop done\|ticks per ns" >> $F; done
bash tools/ab_env.sh "three_sets:" "two_sets:VPD_LIB_PATH=$R/tools/probe/ab/libsets2.so" >> $F 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "three_sets_512:" "two_sets_512:VPD_LIB_PATH=$R/tools/probe/ab/libsets2.so" >> $F 2>&1
cut -c1-250 $F
