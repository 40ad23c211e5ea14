set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -k "stream" > $OUT/stream_tests.log 2>&1; tail -5 $OUT/stream_tests.log
grep -q passed $OUT/stream_tests.log && ! grep -q failed $OUT/stream_tests.log || exit 1
for s in 0 0 1; do VPD_CONV1X1_STREAM=$s timeout -k 10 200 python3 tools/step_digest.py --arch resnet50 --steps 3 2>&1 | tail -1; done
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "igemm:VPD_CONV1X1_STREAM=0" "stream:" 2>&1 | cut -c1-400
