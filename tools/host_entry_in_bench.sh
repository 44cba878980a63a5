# the host_entry block of bench.py (after the headline, in its process) and tools/host_entry_probe.py (a fresh process):
# the two have disagreed before (a hand-over one chunk early: better in the probe, much worse inside the bench)
f="--no-cpu-baseline --no-secondary --no-tertiary --no-ceilings --no-share-block --no-reference-case --no-shard-block --no-hashtable-block --no-kernel-roofline --no-pmc"
show='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); h=d["host_entry"]
print("bench", round(d["value"]), round(h["h2d_GBs"],1), [(c,k,round(v["value"]),round(v.get("of_pcie_bound",0),3),round(v.get("of_resident",0),3)) for c in ("configs1","configs2") for k,v in h[c].items() if isinstance(v,dict)])'
showp='import sys,json
d=json.loads(sys.stdin.read()); b=d["pcie_bound_segments_per_s"]
print("probe", d["config"], round(d["resident"]["segments_per_s"]), round(d["host_pageable"]["segments_per_s"]), round(d["host_pageable"]["segments_per_s"]/b,3), round(d["host_pinned"]["segments_per_s"]), round(d["host_pinned"]["segments_per_s"]/b,3))'
for i in 1 2; do
  python bench.py --steps 4 --warmup 1 --min-seconds 1 $f 2>/dev/null | python -c "$show"
  python tools/host_entry_probe.py 2 2>/dev/null | tail -1 | python -c "$showp"
done
python tools/host_entry_probe.py 3 2>/dev/null | tail -1 | python -c "$showp"
