set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "wgrad128 or r50 or resnet50 or wr50 or reference_gradients" 2>&1 | tail -3
python3 tools/step_digest.py 2>/dev/null | tail -1
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "tco128:VPD_WG2_TCO256=0" "tco256:" 2>&1 | cut -c1-330
