#!/bin/bash
# Per-kernel durations of the isolated fp32 launch sets (tools/valu_probe.py) under rocprofv3 --kernel-trace.
# usage (on the GPU box): tools/kprobe.sh <outdir> [segments] [signals]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python tools/valu_probe.py ${2:-2048} ${3:-1} 5 > $O/probe.log 2>$O/probe.err
python tools/profile_summary.py $(ls $O/raw/*/*kernel_trace.csv) > $O/by_grid.csv
rm -rf $O/raw
grep -v "at::\|elementwise\|calib" $O/by_grid.csv | head -30; cat $O/probe.log
