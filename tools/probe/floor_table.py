#!/usr/bin/env python3
"""rocprofv3 --kernel-trace view of tools/probe/floor_probe --noevents --reps R:
   floor_table.py <kernel_trace.csv> R
The probe launches (empty256, writer<F>(B), successor) triples in a fixed loop order (successor kind, MB, flavour, rep);
this pairs the trace rows with that order and prints the median duration of writer and successor and the gap between them."""
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if any(k in r["Kernel_Name"] for k in ("writer<", "empty_kernel", "finalize_kernel", "stream_kernel"))]
R = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
FN, SN, MBS = ["plain", "sc1", "sc0sc1", "nt"], ["empty1", "empty256", "finalize", "stream"], [0, 4, 16, 32, 128]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
i = 0
print("%-8s %6s %-9s %10s %9s %8s   (rocprofv3 --kernel-trace, medians, us)" % ("flavour", "MB", "successor", "writer_us", "succ_us", "gap_us"))
for kind in range(4):
    for mb in MBS:
        for f in range(4):
            if mb == 0 and f: continue
            w, s, g = [], [], []
            for rep in range(R + 3):
                q, wr, su = rows[i:i + 3]; i += 3
                assert "empty_kernel" in q["Kernel_Name"] and "writer<" in wr["Kernel_Name"], (i, q["Kernel_Name"], wr["Kernel_Name"])
                if rep < 3: continue
                w.append(dur(wr)); s.append(dur(su)); g.append((int(su["Start_Timestamp"]) - int(wr["End_Timestamp"])) / 1e3)
            print("%-8s %6d %-9s %10.2f %9.2f %8.2f" % (FN[f], mb, SN[kind], statistics.median(w), statistics.median(s), statistics.median(g)))
rest = rows[i:]
for nb in (0, 1):
    ch = rest[nb * 2000:(nb + 1) * 2000]
    if len(ch) == 2000:
        wall = (int(ch[-1]["End_Timestamp"]) - int(ch[0]["Start_Timestamp"])) / 1e3 / 2000
        print("# chain of 2000 empty kernels (%s): %.2f us per launch wall, median duration %.2f us" % ("1 block" if nb == 0 else "256 blocks", wall, statistics.median(dur(r) for r in ch)))
