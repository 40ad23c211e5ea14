#!/usr/bin/env python3
"""One training step as a timeline: every kernel launch of the LAST profiled step in order, with its duration and the gap
to its predecessor, from a rocprofv3 --kernel-trace csv.   tools/step_timeline.py <kernel_trace.csv> [launches_per_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the zero_ranges launch that precedes pack_input
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("pack_input") or "pack_input_kernel" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("need at least two steps in the trace")
a, b = starts[-2] - 1, starts[-1] - 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")
    print("%8.1f  +%5.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:90]))
    prev_end = e
    tot += (e - s) / 1e3
print("launches %d, kernel time %.1f us, wall %.1f us" % (len(step), tot, (prev_end - t0) / 1e3))
