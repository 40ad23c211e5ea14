set -u
R=$PWD; OUT=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $OUT/r06_gputests_mid.log 2>&1; echo "tests rc $?"; tail -15 $OUT/r06_gputests_mid.log | cut -c1-300
