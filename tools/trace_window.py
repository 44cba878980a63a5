#!/usr/bin/env python3
"""Flat timeline (start, duration, stream, grid y, kernel) of the library's kernels in a slice of a rocprofv3
kernel trace: trace_window.py <kernel_trace.csv> [from_fraction] [to_fraction] [max_rows]."""
import csv, sys
csv.field_size_limit(1 << 30)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wspr" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.97
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 700
sub = rows[int(len(rows) * lo):int(len(rows) * hi)][:cap]
t0 = int(sub[0]["Start_Timestamp"])
for r in sub:
    nm = r["Kernel_Name"].split("wspr::(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")[:40]
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f s%s y%s %s" % ((b - t0) / 1e3, (e - b) / 1e3, r["Stream_Id"], r["Grid_Size_Y"], nm))
