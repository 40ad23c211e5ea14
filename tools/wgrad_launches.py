#!/usr/bin/env python3
"""Per-launch durations of the weight-gradient kernels from a rocprofv3 --kernel-trace csv (last step's launches)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 9
for r in sel[-per:]:
    print("%-60s %8.1f us  grid %s" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?")))
