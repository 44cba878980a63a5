#!/bin/bash
# VALU instructions issued per kernel over a few configs[2] steps (rocprofv3 --pmc SQ_INSTS_VALU): which kernel the
# step's vector issue slots go to.  usage (GPU box): tools/insts_by_kernel.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/raw -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --min-seconds 0 --inflight 1 > $O/insts_bench.json 2> $O/insts.err
python - $(ls $O/raw/*/*counter_collection.csv) > $O/insts_by_kernel.txt <<'PY'
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = name.split("wspr::(anonymous namespace)::")[-1].split("(")[0][:44] if "wspr::" in name else "other:" + name[:30]
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[short] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
print("kernel,dispatches,insts_valu,share,valu_active_share_of_wave_cycles,wait_any_share")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:24]:
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print("%s,%d,%.4g,%.4f,%.3f,%.3f" % (k, n[k], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / tot, v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_WAIT_ANY"] / wc))
print("total,,%.4g" % tot)
PY
rm -rf $O/raw
cat $O/insts_by_kernel.txt
