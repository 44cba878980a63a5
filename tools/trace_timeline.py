#!/usr/bin/env python3
"""Per-stream timeline of one bench step from a rocprofv3 kernel trace: busy time, idle gaps and
kernel list per HIP stream, and how many streams are busy on average (GPU concurrency)."""
import csv, sys, collections
csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), short, int(r["Grid_Size_Y"])))
rows.sort()
# steps: fft_bank launches with seg_list == all -> find the periodic pattern: take the last 40 % of the trace
t_lo = rows[int(len(rows) * float(sys.argv[2]) if len(sys.argv) > 2 else int(len(rows) * 0.6))][0]
t_hi = t_lo + int(float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 7e6)
win = [r for r in rows if t_lo <= r[0] < t_hi]
streams = collections.defaultdict(list)
for r in win:
    streams[r[2]].append(r)
print("window %.3f ms, %d kernels, streams: %s" % ((t_hi - t_lo) / 1e6, len(win), sorted(streams)))
for s, rs in sorted(streams.items()):
    busy = sum(e - b for b, e, *_ in rs)
    print("--- stream %d: %d kernels, busy %.3f ms" % (s, len(rs), busy / 1e6))
    prev = None
    for b, e, _, n, gy in rs:
        gap = (b - prev) / 1e3 if prev else 0.0
        print("  +%8.1f us  gap %7.1f  dur %7.1f  %s [y=%d]" % ((b - t_lo) / 1e3, gap, (e - b) / 1e3, n, gy))
        prev = e
