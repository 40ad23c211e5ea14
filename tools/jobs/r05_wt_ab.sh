# every 16-byte epilogue / BatchNorm / stem-pool store as a write-through (sc1) or nt store against plain stores, same box.
set -u
R=$PWD; OUT=$R/gpurun_out
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r05_wt_ab.txt 2>&1
bash tools/ab_env.sh "plain:" "sc1:VPD_LIB_PATH=$R/tools/probe/ab/libwt1.so" "nt:VPD_LIB_PATH=$R/tools/probe/ab/libwt3.so" "r04:VPD_LIB_PATH=$R/tools/probe/ab/libr04.so" >> $OUT/r05_wt_ab.txt 2>&1
cat $OUT/r05_wt_ab.txt
