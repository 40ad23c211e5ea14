set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/r04_gputests.log 2>&1; tail -2 $OUT/r04_gputests.log
