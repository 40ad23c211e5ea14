set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_mask_probe.txt
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $F
export BENCH_PWS_LAYERS=l2,l3,l4
for k in 1 2 3; do
python3 tools/bench_pws.py 256 normal 2>&1 | grep -v amdgpu >> $F
VPD_LIB_PATH=$R/tools/probe/ab/libmaskprobe.so python3 tools/bench_pws.py 256 halo_lanes_one_third_masked 2>&1 | grep -v amdgpu >> $F
done
cat $F
