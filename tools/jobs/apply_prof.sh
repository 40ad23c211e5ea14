set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ap -o p -- python3 $R/tools/bench_apply.py --batches 30 > /dev/null 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/ap/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s calls %6s avg %8.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
rm -rf $OUT/ap
