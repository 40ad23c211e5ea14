#!/usr/bin/env python3
"""One training step as a timeline: every kernel launch of one profiled step (the quietest of the last ten) in order, with its duration and the gap
to its predecessor, from a rocprofv3 --kernel-trace csv.   tools/step_timeline.py <kernel_trace.csv> [launches_per_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the zero_ranges launch that precedes pack_input
starts = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("need at least two steps in the trace")
# the step with the smallest sum of inter-dispatch gaps among the last ten complete ones (one host hiccup under the profiler
# puts 40-120 us gaps into a single step; the per-step statistics over all steps are in tools/launch_gaps.py's output)
best = None
for k in range(max(1, len(starts) - 10), len(starts)):
    a, b = starts[k - 1] - 1, starts[k] - 1
    cand = rows[a:b]
    gaps = sum(max(0, int(cand[i]["Start_Timestamp"]) - int(cand[i - 1]["End_Timestamp"])) for i in range(1, len(cand)))
    if best is None or gaps < best[0]:
        best = (gaps, cand)
step = best[1]
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
