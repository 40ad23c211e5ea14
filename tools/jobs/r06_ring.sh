# With the geo K loop the issue order no longer hides it: is the K loop bound by the LDS-DMA bytes in flight?  Ring depth NS 6 on the 256 x 64 tile
# (5 shipped), NS 9 on the 128 x 64 tile (7 shipped); VPD_LIB_PATH variants, alternating; digests equal by construction (same K order).
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_ring_depth.txt
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $F
( echo "digest tree:"; python3 tools/step_digest.py 2>/dev/null; echo "digest NS 6 / 9:"; VPD_LIB_PATH=$R/tools/probe/ab/libns6_9.so python3 tools/step_digest.py 2>/dev/null ) >> $F 2>&1
bash tools/ab_env.sh "ns5_7:" "ns6_7:VPD_LIB_PATH=$R/tools/probe/ab/libns6.so" "ns5_9:VPD_LIB_PATH=$R/tools/probe/ab/libns_9.so" "ns6_9:VPD_LIB_PATH=$R/tools/probe/ab/libns6_9.so" >> $F 2>&1
cut -c1-200 $F
