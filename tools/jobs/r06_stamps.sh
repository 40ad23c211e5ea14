set -u
R=$PWD; OUT=$R/gpurun_out
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r06_geo_stamps.txt
for L in l3 l2 l4; do
  export BENCH_PWS_LAYERS=$L
  echo "=== $L geo" >> $OUT/r06_geo_stamps.txt
  VPD_LIB_PATH=$R/tools/probe/ab/libstamps.so python3 tools/bench_pws.py 256 geo >> $OUT/r06_geo_stamps.txt 2>&1
  echo "=== $L generic" >> $OUT/r06_geo_stamps.txt
  VPD_PWS_GEO=0 VPD_LIB_PATH=$R/tools/probe/ab/libstamps.so python3 tools/bench_pws.py 256 generic >> $OUT/r06_geo_stamps.txt 2>&1
  echo "=== $L geo, reads issued but not waited for" >> $OUT/r06_geo_stamps.txt
  VPD_LIB_PATH=$R/tools/probe/ab/libstampsdummy.so python3 tools/bench_pws.py 256 geo_unwaited >> $OUT/r06_geo_stamps.txt 2>&1
done
grep -v amdgpu.ids $OUT/r06_geo_stamps.txt | grep -v "L origin\|L entry\|first entry\|stats flushed\|all epilogues"
