set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for lib in "" "$R/tools/probe/ab/libold.so"; do VPD_LIB_PATH=$lib timeout -k 10 300 python tools/step_digest.py 2>&1 | tail -1 | sed "s#^#lib=${lib:-tree} #"; done | tee $OUT/pool_digest.txt
bash tools/prof_ab.sh pool "new:VPD_X=1" "old:VPD_LIB_PATH=$R/tools/probe/ab/libold.so" && python3 - <<PY
import csv
def load(f):
    return {r["Name"]: (float(r["Calls"])/25, float(r["TotalDurationNs"])/25e3) for r in csv.DictReader(open(f))}
a=load("gpurun_out/pool_new_kernel_stats.csv"); b=load("gpurun_out/pool_old_kernel_stats.csv")
for n in sorted(set(a)|set(b), key=lambda n: -(b.get(n,(0,0))[1])):
    ca,ua=a.get(n,(0,0)); cb,ub=b.get(n,(0,0))
    if abs(ua-ub)>1.5 or ca!=cb: print("%-60s new %5.1f x %8.1f us | old %5.1f x %8.1f us | %+7.1f" % (n[:60],ca,ua,cb,ub,ua-ub))
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
