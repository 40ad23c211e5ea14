# Round-4 evidence, second pass (after the transforming loaders went in): the whole r04 set again from ONE box, the XF on / off
# table from the same box, the stamps of the XF kernels, and the GPU test suite on the final tree.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/jobs/r04_evidence.sh 2>&1 | tee $OUT/r04_evidence_lines.txt
bash tools/ab_env.sh "xf:VPD_CONV_XF=1" "base:VPD_CONV_XF=0" > $OUT/r04_xf_ab_256.txt 2>&1
for v in 1 0; do VPD_CONV_XF=$v timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s/^/VPD_CONV_XF=$v /"; done > $OUT/r04_xf_digest.txt
VPD_LIB_PATH=$R/tools/probe/ab/libstamps.so timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2> $OUT/r04_xf_stamps.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/r04_gputests.log 2>&1; tail -2 $OUT/r04_gputests.log
