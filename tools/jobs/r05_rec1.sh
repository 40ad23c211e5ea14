set -u
R=$PWD; OUT=$R/gpurun_out
for s in 0 1; do VPD_BNECK_RECOMPUTE=$s timeout -k 10 200 python3 tools/step_digest.py --arch resnet50 --steps 3 2>&1 | tail -1; done
for s in 0 1; do VPD_BNECK_RECOMPUTE=$s timeout -k 10 200 python3 tools/step_digest.py --arch resnet50 --steps 2 --batch 64 2>&1 | tail -1; done
AB_EXTRA="--arch resnet50" bash tools/ab_env.sh "store:VPD_BNECK_RECOMPUTE=0" "recompute:" 2>&1 | cut -c1-330
