# full GPU suite (timed), whole-step digests of the tree and of round 4's library, same-box A/B
set -u
R=$PWD; OUT=$R/gpurun_out; TAG=${1:-r05_suite}; L4=$R/tools/probe/ab/libr04.so
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/$TAG.txt 2>&1
( echo "digest tree:"; python3 tools/step_digest.py 2>/dev/null; echo "digest r04:"; VPD_LIB_PATH=$L4 python3 tools/step_digest.py 2>/dev/null; echo "digest tree resnet50:"; python3 tools/step_digest.py --arch resnet50 --batch 64 2>/dev/null; echo "digest r04 resnet50:"; VPD_LIB_PATH=$L4 python3 tools/step_digest.py --arch resnet50 --batch 64 2>/dev/null ) >> $OUT/$TAG.txt 2>&1
python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_gputests.log 2>&1; tail -2 $OUT/${TAG}_gputests.log >> $OUT/$TAG.txt
bash tools/ab_env.sh "new:" "r04:VPD_LIB_PATH=$L4" >> $OUT/$TAG.txt 2>&1
cat $OUT/$TAG.txt
