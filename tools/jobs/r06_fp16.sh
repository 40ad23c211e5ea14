set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k fp16 > $OUT/r06_fp16_tests.log 2>&1; echo "tests rc $?"; tail -25 $OUT/r06_fp16_tests.log
cat $OUT/parity_fp16_*.json
for k in 1 2; do
python3 tools/bench_apply.py --batches 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', {k: (round(v) if isinstance(v,(int,float)) else v) for k,v in d.items() if k in ('forward_resident','loop_host_u8','graph_only')})"
python3 tools/bench_apply.py --batches 30 --dtype fp16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16', {k: (round(v) if isinstance(v,(int,float)) else v) for k,v in d.items() if k in ('forward_resident','loop_host_u8','graph_only')})"
done
