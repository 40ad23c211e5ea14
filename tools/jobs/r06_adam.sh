# Optimizer step inside the backward pass (vpd_plan_arm_adamw, VPD_EARLY_ADAMW=1 default): digests must equal the serial step's, then alternating runs.
set -u
R=$PWD; OUT=$R/gpurun_out; F=$OUT/r06_ab_early_adamw.txt
( echo "digest AdamW inside the backward pass (default):"; python3 tools/step_digest.py 2>&1 | tail -1; echo "digest VPD_EARLY_ADAMW=0:"; VPD_EARLY_ADAMW=0 python3 tools/step_digest.py 2>&1 | tail -1 ) > $F 2>&1
bash tools/ab_env.sh "early_adamw:" "serial_adamw:VPD_EARLY_ADAMW=0" >> $F 2>&1
AB_EXTRA="--batch 512" bash tools/ab_env.sh "early_adamw_512:" "serial_adamw_512:VPD_EARLY_ADAMW=0" >> $F 2>&1
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "early_adamw_r50:" "serial_adamw_r50:VPD_EARLY_ADAMW=0" >> $F 2>&1
cut -c1-100 $F
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_fp16_gpu.py -q -m gpu -x -k "matches_reference_and_oracle or loss_scaler or lazy or optimizer or adamw" > $OUT/r06_adam_tests.log 2>&1; tail -3 $OUT/r06_adam_tests.log
