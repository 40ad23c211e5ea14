set -u
R=$PWD; OUT=$R/gpurun_out
for v in 0 1; do VPD_STREAM_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "stream" 2>&1 | tail -1; done
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "v0:VPD_STREAM_VARIANT=0" "v1:VPD_STREAM_VARIANT=1" 2>&1 | cut -c1-100
