set -u
R=$PWD; OUT=$R/gpurun_out
mkdir -p $OUT/f16probe $OUT/bf16probe
VPD_F16_TRAIN_PROBE=1 VPD_FORCE_DTYPE_PROBE=fp16 timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "matches_reference_and_oracle" > $OUT/r06_f16probe.log 2>&1; tail -30 $OUT/r06_f16probe.log | cut -c1-300
cp $OUT/parity_r*.json $OUT/parity_wr*.json $OUT/parity_c1*.json $OUT/f16probe/ 2>/dev/null
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "matches_reference_and_oracle" > $OUT/r06_bf16probe.log 2>&1; tail -3 $OUT/r06_bf16probe.log
cp $OUT/parity_r*.json $OUT/parity_wr*.json $OUT/parity_c1*.json $OUT/bf16probe/ 2>/dev/null
