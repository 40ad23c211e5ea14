# CU partition.  VPD_RESERVE_CUS=R sizes every persistent grid of the step to 256 - R CUs; VPD_WG_OVERLAP=1 VPD_WG_CUMASK=R
# runs the grouped weight gradients on a side stream confined to R CUs (grids sized to R).  Digests first (must not change), then alternating runs.
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_wgrad_overlap2.txt
( echo "digest default:"; python3 tools/step_digest.py 2>/dev/null
  echo "digest VPD_RESERVE_CUS=64:"; VPD_RESERVE_CUS=64 python3 tools/step_digest.py 2>/dev/null
  echo "digest VPD_RESERVE_CUS=64 VPD_WG_OVERLAP=1 VPD_WG_CUMASK=64:"; VPD_RESERVE_CUS=64 VPD_WG_OVERLAP=1 VPD_WG_CUMASK=64 python3 tools/step_digest.py 2>/dev/null ) > $F 2>&1
bash tools/ab_env.sh "serial:" "reserve16:VPD_RESERVE_CUS=16" "reserve32:VPD_RESERVE_CUS=32" "reserve64:VPD_RESERVE_CUS=64" \
  "split192_64:VPD_RESERVE_CUS=64,VPD_WG_OVERLAP=1,VPD_WG_CUMASK=64" "split160_96:VPD_RESERVE_CUS=96,VPD_WG_OVERLAP=1,VPD_WG_CUMASK=96" \
  "split128_128:VPD_RESERVE_CUS=128,VPD_WG_OVERLAP=1,VPD_WG_CUMASK=128" >> $F 2>&1
cut -c1-110 $F
VPD_RESERVE_CUS=16 timeout -k 10 600 python -m pytest tests/test_pws_gpu.py tests/test_ddp_gpu.py -q -m gpu -k "pws or grid_barrier" > $OUT/r06_reserve_tests.log 2>&1; tail -3 $OUT/r06_reserve_tests.log
