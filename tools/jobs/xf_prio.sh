set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "loaders" 2>&1 | tail -2
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" "noprio:VPD_XF_ABLATE=8" "prioall:VPD_XF_ABLATE=16" > $OUT/xf_prio.txt 2>&1
cut -c1-150 $OUT/xf_prio.txt
bash tools/jobs/xf_stamps.sh | grep -A 15 "xf stamps <256"
