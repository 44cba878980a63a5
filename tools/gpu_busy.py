#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: fraction of the time at least one kernel is running, average number of
kernels running, and per-kernel totals inside the analysed window (the last 40 % of the trace by default)."""
import csv, sys, collections
csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
lo = rows[int(len(rows) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.6))][0]
hi = rows[int(len(rows) * (float(sys.argv[3]) if len(sys.argv) > 3 else 0.95))][0]
ev = []
tot = collections.Counter()
for b, e, n in rows:
    b2, e2 = max(b, lo), min(e, hi)
    if e2 > b2:
        ev.append((b2, 1)); ev.append((e2, -1)); tot[n] += e2 - b2
ev.sort()
busy = 0; area = 0; depth = 0; prev = lo
for t, d in ev:
    if depth > 0: busy += t - prev
    area += depth * (t - prev)
    depth += d; prev = t
print("window %.2f ms: GPU busy %.1f %%, mean kernels in flight while busy %.2f" % ((hi - lo) / 1e6, 100.0 * busy / (hi - lo), area / max(busy, 1)))
for n, v in tot.most_common(14):
    print("  %-40s %8.2f ms  (%.1f %% of the window)" % (n[:40], v / 1e6, 100.0 * v / (hi - lo)))
