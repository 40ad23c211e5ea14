# grouped weight gradients + slab reduces + bucket hand-over on a side stream behind per-stage events (VPD_WG_OVERLAP=1),
# optionally confined to n CUs (VPD_WG_CUMASK=n, hipExtStreamCreateWithCUMask), with the layer3 / layer4 launches merged or not.
set -u
R=$PWD; OUT=$R/gpurun_out
( echo "digest default:"; python3 tools/step_digest.py 2>/dev/null; echo "digest VPD_WG_OVERLAP=1:"; VPD_WG_OVERLAP=1 python3 tools/step_digest.py 2>/dev/null;
  echo "digest VPD_WG_OVERLAP=1 VPD_WG_CUMASK=64:"; VPD_WG_OVERLAP=1 VPD_WG_CUMASK=64 python3 tools/step_digest.py 2>/dev/null ) > $OUT/r06_ab_wgrad_overlap.txt 2>&1
bash tools/ab_env.sh "serial:" "overlap:VPD_WG_OVERLAP=1" "overlap_unmerged:VPD_WG_OVERLAP=1,VPD_WG_MERGE=0" "overlap_mask64:VPD_WG_OVERLAP=1,VPD_WG_CUMASK=64" "overlap_mask96:VPD_WG_OVERLAP=1,VPD_WG_CUMASK=96" "overlap_mask128:VPD_WG_OVERLAP=1,VPD_WG_CUMASK=128" >> $OUT/r06_ab_wgrad_overlap.txt 2>&1
cut -c1-100 $OUT/r06_ab_wgrad_overlap.txt
