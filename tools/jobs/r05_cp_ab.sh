# cache-policy variants of the big streams, same box (libs built by tools/build_variant_lib.sh <name> <flags>); usage: r05_cp_ab.sh tag name1 name2 ...
set -u
R=$PWD; OUT=$R/gpurun_out; TAG=$1; shift
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/${TAG}.txt 2>&1
CFG="plain:"
for n in "$@"; do CFG="$CFG $n:VPD_LIB_PATH=$R/tools/probe/ab/lib$n.so"; done
bash tools/ab_env.sh $CFG >> $OUT/${TAG}.txt 2>&1
cat $OUT/${TAG}.txt
