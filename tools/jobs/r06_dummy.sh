set -u
R=$PWD; OUT=$R/gpurun_out
export BENCH_PWS_LAYERS=l2,l3,l4
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r06_geo_dummy.txt
for k in 1 2; do
python3 tools/bench_pws.py 256 geo >> $OUT/r06_geo_dummy.txt 2>&1
VPD_PWS_GEO=0 python3 tools/bench_pws.py 256 generic >> $OUT/r06_geo_dummy.txt 2>&1
VPD_LIB_PATH=$R/tools/probe/ab/libgeodummy.so python3 tools/bench_pws.py 256 geo_reads_unwaited >> $OUT/r06_geo_dummy.txt 2>&1
done
grep -v amdgpu.ids $OUT/r06_geo_dummy.txt
