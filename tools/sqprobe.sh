#!/bin/bash
# SQ counters of the isolated fp32 launch sets (tools/valu_probe.py): where the waves of each kernel spend their cycles.
# usage (on the GPU box): tools/sqprobe.sh <outdir> [segments] [signals]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/raw -- python tools/valu_probe.py ${2:-2048} ${3:-1} 3 > $O/sq_probe.log 2>$O/sq_probe.err
python tools/sq_summarise.py $(ls $O/raw/*/*counter_collection.csv) > $O/sq_summary.json
rm -rf $O/raw
python - $O/sq_summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if "SQ_WAVE_CYCLES" not in v: continue
    print("%-28s valu %.2f lds %.2f wait_any %.2f wait_inst %.2f bankconf/ldsactive %.2f insts_valu %.3g" % (k[:28], v.get("SQ_ACTIVE_INST_VALU_share_of_wave_cycles",0), v.get("SQ_ACTIVE_INST_LDS_share_of_wave_cycles",0), v.get("SQ_WAIT_ANY_share_of_wave_cycles",0), v.get("SQ_WAIT_INST_ANY_share_of_wave_cycles",0), v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_ACTIVE_INST_LDS",1)), v.get("SQ_INSTS_VALU",0)))
PY
