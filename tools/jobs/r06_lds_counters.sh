# Round 6, item 1: what bounds the K loop of the dominant 3x3 kernel?  SQ counters per kernel name, counters only (no trace domains),
# the program itself behind `--`.  One pass per counter group (8 SQ slots); a pass with an unknown counter name fails alone.
set -u
R=$PWD; OUT=$R/gpurun_out; D=$OUT/r06_pmc
mkdir -p $D
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r06_lds_counters.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/r06_counter_list.txt 2>&1
grep -o 'SQ_LDS[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*' $OUT/r06_counter_list.txt | sort -u > $OUT/r06_counter_names.txt
PARGS="--steps 4 --warmup 2 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply --no-parity"
pass() {   # pass <name> <counters...>
  local n=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" -d $D/$n -o pmc --output-format csv -- python3 $R/bench.py $PARGS > $D/$n.log 2>&1
  echo "pass $n ($*): rc $?" >> $OUT/r06_lds_counters.txt
}
pass lds   SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass wait  SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass lds2  SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass inst  SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass coex  SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
PMC_SQ_JSON=$OUT/r06_lds_counters.json python3 $R/tools/pmc_sq.py $D conv3x3_pws_kernel conv3x3_c64_persistent conv_wgrad128_persistent conv1x1_ws_kernel conv_stem_persistent conv_wgrad_halo >> $OUT/r06_lds_counters.txt 2>&1
for n in lds wait lds2 inst coex; do tail -3 $D/$n.log > $OUT/r06_pmc_$n.tail; done
find $D -name '*.csv' -size +20M -delete
head -40 $OUT/r06_lds_counters.txt
