# FETCH_SIZE / WRITE_SIZE passes + kernel stats of the ResNet-50 bench -> bytes per launch and achieved TB/s per kernel
set -u
R=$PWD; OUT=$R/gpurun_out; TAG=r05_resnet50
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))"
cd /tmp && export TMPDIR=/tmp
PARGS="--arch resnet50 --steps 4 --warmup 2 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply"
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o pmc --output-format csv -- python3 $R/bench.py $PARGS > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python3 $R/bench.py --arch resnet50 --steps 10 --warmup 3 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply > /dev/null 2>&1
python3 $R/tools/bandwidth_table.py $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_prof/p_kernel_stats.csv 8 > $OUT/${TAG}_bandwidth.txt 2>&1
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_prof
head -40 $OUT/${TAG}_bandwidth.txt
