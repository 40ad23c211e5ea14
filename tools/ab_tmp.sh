timeout 900 python -m pytest tests/test_model_gpu.py tests/test_apply_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -1
for i in 1 2 3; do
for Q in 0 1; do
VPD_STEM_PAIR=$Q timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair $Q', round(d['value']))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o p --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --steps 10 --warmup 3 --profile-steps 0 > /dev/null 2>&1
grep -h "stem_pool" $(find /tmp/prof_q -name "*kernel_stats.csv") | cut -c1-110
