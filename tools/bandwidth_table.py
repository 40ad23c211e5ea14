#!/usr/bin/env python3
"""Achieved fabric-side bandwidth per kernel: bytes per launch from tools/pmc_traffic.py's JSON divided by the average
duration in a rocprofv3 --kernel-trace --stats CSV of the same build.
usage: bandwidth_table.py <pmc_traffic.json> <kernel_stats.csv> [min_MB]"""
import csv
import json
import sys

traffic = json.load(open(sys.argv[1]))["kernels"]
min_mb = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
    dur["adamw_pack_kernel" if n.startswith("_Z17adamw_pack_kernel") else n] = float(r["AverageNs"]) / 1e3
print("# fabric-side bytes (2*FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) / rocprofv3 average duration")
print("%-58s %10s %9s %8s" % ("kernel", "MB/launch", "avg us", "TB/s"))
rows = []
for name, t in traffic.items():
    key = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if key.startswith("_Z17adamw_pack_kernel"):
        key = "adamw_pack_kernel"
    if key not in dur:
        continue
    b = (2.0 * t["fetch_KB_per_launch"] + t["write_KB_per_launch"]) * 1024.0      # KB -> bytes (guide: FETCH_SIZE x2 on gfx950)
    if b / 1e6 < min_mb:
        continue
    rows.append((b / 1e6 * dur[key], key, b / 1e6, dur[key], b / dur[key] / 1e6))
for _, key, mb, us, tbs in sorted(rows, reverse=True):
    print("%-58s %10.1f %9.1f %8.2f" % (key[:58], mb, us, tbs))
