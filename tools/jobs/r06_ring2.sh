set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_ring_depth2.txt
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $F
bash tools/ab_env.sh "ns7:" "ns8:VPD_LIB_PATH=$R/tools/probe/ab/libns_8.so" "ns9:VPD_LIB_PATH=$R/tools/probe/ab/libns_9.so" "ns10:VPD_LIB_PATH=$R/tools/probe/ab/libns_10.so" >> $F 2>&1
cut -c1-130 $F
