#!/usr/bin/env python3
"""Timeline of one decode step from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration and the idle gap before it.
usage: step_timeline.py kernel_trace.csv [nth_dec_embed=60]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 60
idx = [i for i, r in enumerate(rows) if "k_dec_embed" in r["Kernel_Name"]]
a, b = idx[nth], idx[nth + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; tot = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    print("%9.2f us  dur %7.2f  gap %6.2f  %-40s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, r.get("Grid_Size_X", r.get("Grid_Size"))))
    tot += e - s; gaps += max(0, s - prev_end); prev_end = e
print("step: %d kernels, busy %.1f us, gaps %.1f us, span %.1f us" % (b - a, tot / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
