#!/usr/bin/env python3
"""Compact per-kernel summary of a rocprofv3 --kernel-trace --stats run of bench.py.
usage: prof_summary.py <kernel_stats.csv> <steps_profiled>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-64s %7s %9s %8s %6s" % ("kernel", "calls/st", "ms/step", "avg_us", "%"))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-64s %7.1f %9.3f %8.1f %6.1f" % (r["Name"][:64], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6,
                                            float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel ms/step %.3f" % (tot / steps / 1e6))
