#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_k1.py.
usage: pmc_summarise.py fetch.csv write.csv [segments]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  They are calibrated on the
stream-copy kernel of known size launched by the same script (same 4-byte-per-lane access
width), as MI355X_MICROARCH.md section HBM prescribes, then applied to K1/K2/K3."""
import csv, json, sys, collections
csv.field_size_limit(1 << 30)

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = ("calib_copy" if "calib_copy" in name else "fft_bank_avg" if "fft_bank_avg" in name else
               "fft_bank" if "fft_bank" in name else "time_average" if "time_average" in name else
               "pick_peaks" if "pick_peaks" in name else "coarse_sync" if "coarse_sync" in name else None)
        if key:
            acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
NSEG = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
copy_bytes = 4 * (1 << 28)
cal_r = copy_bytes / (fetch["calib_copy"] * 1024.0)
cal_w = copy_bytes / (write["calib_copy"] * 1024.0)
K1 = 360000 + 4 * 417 * 347
alg = {"fft_bank": K1 * NSEG, "fft_bank_avg": K1 * NSEG, "time_average": 4 * 417 * 347 * NSEG, "pick_peaks": None,
       "coarse_sync": None}
out = {"counter_unit": "KiB per dispatch", "segments": NSEG, "calibration": {
    "kernel": "calib_copy_kernel, 1 GiB read + 1 GiB written, 4 B per lane",
    "FETCH_SIZE_KiB": fetch["calib_copy"], "WRITE_SIZE_KiB": write["calib_copy"],
    "true_bytes_per_counted_read_byte": cal_r, "true_bytes_per_counted_written_byte": cal_w}, "kernels": {}}
for k in ("fft_bank_avg", "fft_bank", "time_average", "pick_peaks", "coarse_sync"):
    if k not in fetch:
        continue
    rb = fetch[k] * 1024.0 * cal_r
    wb = write[k] * 1024.0 * cal_w
    out["kernels"][k] = {"dispatches_averaged": nf[k], "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write[k],
                         "hbm_read_bytes": rb, "hbm_written_bytes": wb, "hbm_bytes_per_launch": rb + wb,
                         "algorithmic_bytes_per_launch": alg[k],
                         "traffic_over_algorithmic": (rb + wb) / alg[k] if alg[k] else None}
dom = "fft_bank_avg" if "fft_bank_avg" in out["kernels"] else "fft_bank"
out["dominant_kernel"] = dom
out["hbm_bytes_per_launch"] = out["kernels"][dom]["hbm_bytes_per_launch"]
out["hbm_bytes_per_segment"] = out["hbm_bytes_per_launch"] / NSEG
print(json.dumps(out, indent=1))
