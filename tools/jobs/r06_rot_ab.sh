# Experiment: do the blocks of a 3x3 launch camp on a subset of the L2 channels (all of them read 64-channel chunk cc of 512-byte pixels at
# the same time)?  VPD_PWS_ROT: 1 = a block starts at chunk lane0 % nchunks, 2 = (lane0 + nt) % nchunks.
set -u
R=$PWD; OUT=$R/gpurun_out
bash tools/ab_env.sh "rot0:" "rot1:VPD_PWS_ROT=1" "rot2:VPD_PWS_ROT=2" > $OUT/r06_ab_rot.txt 2>&1
cut -c1-330 $OUT/r06_ab_rot.txt
