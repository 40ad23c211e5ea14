# Ablations of the generic pipelined 3x3 kernel on the round-6 tree (library built with -DVPD_ENABLE_ABLATE): which stream bounds a launch?
# 1 no weight DMA, 4 no halo DMA, 8 no epilogue, 16 no barriers, 32 no fragment reads, 64 no MFMA
set -u
R=$PWD; OUT=$R/gpurun_out
export VPD_LIB_PATH=$R/tools/probe/ab/libablate.so VPD_PWS_GEO=0 BENCH_PWS_LAYERS=l2,l3,l4
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r06_pws_ablations.txt
for a in 0 1 4 5 8 13 32 64 96 101 104 109 0; do
  VPD_ABLATE=$a python3 tools/bench_pws.py 256 abl_$a >> $OUT/r06_pws_ablations.txt 2>&1
done
cat $OUT/r06_pws_ablations.txt
