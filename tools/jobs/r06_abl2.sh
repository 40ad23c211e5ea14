# Is the geo K loop bound by the LDS-DMA stream?  Loader-side ablations (shared by the generic and the geo kernels): 1 no weight DMA, 4 no halo DMA, 5 neither.
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_geo_ablations.txt
export VPD_LIB_PATH=$R/tools/probe/ab/libablate.so BENCH_PWS_LAYERS=l2,l3,l4
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $F
for k in 1 2; do for a in 0 1 4 5; do
  VPD_ABLATE=$a python3 tools/bench_pws.py 256 geo_abl_$a 2>&1 | grep -v amdgpu >> $F
  VPD_PWS_GEO=0 VPD_ABLATE=$a python3 tools/bench_pws.py 256 generic_abl_$a 2>&1 | grep -v amdgpu >> $F
done; done
cat $F
