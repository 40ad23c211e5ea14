# geo K loop on the eight-wave 256x128 tile and the 128x128 tile (VPD_PWS_GEO=1, default) against four-wave tiles only (=2) and off (=0):
# digests at 512 crops, bench at 512 / 1024 crops, apply.
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_geo8.txt
timeout -k 10 900 python -m pytest tests/test_pws_gpu.py tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $OUT/r06_geo8_tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/r06_geo8_tests.log
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $F
AB_EXTRA="--batch 512" bash tools/ab_env.sh "geo_all_512:" "geo_4wave_512:VPD_PWS_GEO=2" "generic_512:VPD_PWS_GEO=0" >> $F 2>&1
AB_EXTRA="--batch 1024 --steps 50" bash tools/ab_env.sh "geo_all_1024:" "geo_4wave_1024:VPD_PWS_GEO=2" >> $F 2>&1
for k in 1 2; do for m in 1 2 0; do echo -n "apply VPD_PWS_GEO=$m: " >> $F; VPD_PWS_GEO=$m python3 tools/bench_apply.py --batches 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['forward_resident']), round(d['loop_host_u8']))" >> $F; done; done
cut -c1-110 $F
