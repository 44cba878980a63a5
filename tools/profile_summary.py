#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count and average duration.
The FFT bank is launched by the timed steps (several batches in flight: its launches overlap other kernels) and,
after them, alone by the roofline measurement (wspr_bench_fft_sync: the trace's LAST ten launches of the kernel); with
one slot per batch both have the whole-batch grid, so the last ten are listed in a row of their own to be compared
with bench.py's HIP-event figure."""
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
fft = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "wspr::" not in name:
        continue
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0]
    key = (short, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if short.startswith("fft_bank"):
        fft.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, key))
print("kernel,grid_x_threads,grid_y,workgroup,launches,avg_us,min_us,max_us")
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    v = acc[k]
    print("%s,%d,%d,%d,%d,%.2f,%.2f,%.2f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), max(v)))
if len(fft) >= 10:
    big = max(x[2][1] for x in fft)                       # the whole-batch grid (later launches of other sizes: wspr_bench_valu)
    fft = sorted(x for x in fft if x[2][1] == big)
    v = [x[1] for x in fft[-10:]]
    k = fft[-1][2]
    print("%s[the trace's last 10 launches: the roofline measurement],%d,%d,%d,%d,%.2f,%.2f,%.2f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), max(v)))
