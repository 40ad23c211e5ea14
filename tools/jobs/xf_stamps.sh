set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
VPD_LIB_PATH=$R/tools/probe/ab/libstamps.so timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2> $OUT/xf_stamps.txt
grep -A 16 "stamps" $OUT/xf_stamps.txt | grep -v "^--" | head -150
