#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 PMC pass (counters only, no trace domains):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d <out> -o pmc --output-format csv -- \
      python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --profile-steps 0
  python tools/pmc_mfma_util.py <out> profiles/mfma_util.json

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
reports the sum over the 8 XCDs; MI355X_MICROARCH.md, DVFS note: reads high on dispatches shorter than ~0.3 ms, so
this is an upper bound of the chip-level duty cycle for these 25-us kernels).  SQ_VALU_MFMA_BUSY_CYCLES counts 16
cycles per v_mfma_f32_16x16x32_bf16 per SIMD, so utilisation x 2.5 PFLOP/s x (clock / 2.4 GHz) is the achieved rate.
"""
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
            a = acc.setdefault(name, {})
            c = a.setdefault(r["Counter_Name"], [0, 0.0])
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    res = {}
    for k, a in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or "GRBM_GUI_ACTIVE" not in a:
            continue
        n = a["GRBM_GUI_ACTIVE"][0]
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n
        cyc = a["GRBM_GUI_ACTIVE"][1] / n / 8.0
        if busy <= 0:
            continue
        res[k] = {"launches": n, "mfma_busy_cycles_per_launch": busy, "kernel_cycles_per_launch": cyc,
                  "mfma_util": busy / (cyc * 1024.0)}
    json.dump({"note": __doc__.split("utilisation =")[1].strip(), "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:16]:
        print("%-64s %5d  util %.3f  cycles %8.0f" % (k[:64], v["launches"], v["mfma_util"], v["kernel_cycles_per_launch"]))


if __name__ == "__main__":
    main()
