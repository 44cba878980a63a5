#!/usr/bin/env python3
"""Per-kernel averages of an SQ counter pass (rocprofv3 --pmc SQ_... --kernel-trace): VALU/LDS issue
activity of the fp32-bound kernels.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
usage: sq_summarise.py counter_collection.csv"""
import csv, json, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "wspr::" not in name:
        continue
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    d = {c: sum(x) for c, x in v.items()}
    d["dispatches"] = max(len(x) for x in v.values())
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if c in d:
                d[c + "_share_of_wave_cycles"] = d[c] / wc
    out[k] = d
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))), indent=1))
