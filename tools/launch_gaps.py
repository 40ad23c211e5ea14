#!/usr/bin/env python3
"""Inter-dispatch gaps of the train step, from a rocprofv3 --kernel-trace csv of bench.py:
   tools/launch_gaps.py <p_kernel_trace.csv> [skip_steps]
A step runs from one pack_input launch to the next.  Per step: launches, sum of kernel durations, sum of the gaps
start[i+1] - end[i] between consecutive dispatches (negative gaps = overlap, counted as they are), wall time; then the
median over the steps and the distribution of the individual gaps.  This is the GPU-side cost of a dependent kernel
boundary in THIS step (MI355X_MICROARCH.md row "boundary": 1.45-1.9 us) -- not the host's launch rate."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
if len(starts) < skip + 3:
    sys.exit("need more steps in the trace (%d found)" % len(starts))
per_step, gaps_all, by_pred = [], [], {}
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    step = rows[a:b]
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
    gaps = [(int(n["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3 for p, n in zip(step[:-1], step[1:])]
    # the gap into the next step's first launch belongs to this step (host work between steps shows up here)
    tail = (int(rows[b]["Start_Timestamp"]) - int(step[-1]["End_Timestamp"])) / 1e3
    wall = (int(rows[b]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    per_step.append((len(step), dur, sum(gaps), tail, wall))
    gaps_all += gaps
    for p, g in zip(step[:-1], gaps):
        by_pred.setdefault(p["Kernel_Name"].replace("void ", "").split("(")[0][:48], []).append(g)
med = lambda k: statistics.median(s[k] for s in per_step)
print("steps analysed: %d (first %d skipped)" % (len(per_step), skip))
print("launches per step            %8.0f" % med(0))
print("sum of kernel durations      %8.1f us" % med(1))
print("sum of inter-dispatch gaps   %8.1f us   (%.2f us per boundary)" % (med(2), med(2) / max(med(0) - 1, 1)))
print("gap into the next step       %8.1f us" % med(3))
print("step wall (first start -> next step's first start) %8.1f us" % med(4))
g = sorted(gaps_all)
q = lambda f: g[min(len(g) - 1, int(f * len(g)))]
print("individual gaps: min %.2f  p10 %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f us" % (g[0], q(.1), q(.5), q(.9), q(.99), g[-1]))
print("largest mean gap by PREDECESSOR kernel:")
for name, v in sorted(by_pred.items(), key=lambda kv: -statistics.mean(kv[1]))[:12]:
    print("  %-48s n/step %5.1f  mean gap %6.2f us" % (name, len(v) / len(per_step), statistics.mean(v)))
