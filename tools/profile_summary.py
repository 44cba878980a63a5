#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count and average duration.
The FFT bank is launched in two shapes by bench.py: per-slot shares of the batch inside the timed
steps, and whole-batch launches (grid y = segments) by the roofline measurement; this table keeps
them apart so the roofline launches can be compared with bench.py's HIP-event figure."""
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "wspr::" not in name:
        continue
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0]
    key = (short, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,grid_x_threads,grid_y,workgroup,launches,avg_us,min_us,max_us")
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    v = acc[k]
    print("%s,%d,%d,%d,%d,%.2f,%.2f,%.2f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), max(v)))
