#!/bin/bash
# (GPU box) Is the frequency scan bound by its SAMPLE reads?  The launch sets of tools/valu_probe.py over 2 048 candidates
# whose samples are (a) their own (680 MB per frequency-scan launch from HBM: 332 KB per candidate in 128-byte pieces
# 1 KB apart) and (b) those of 8 segments (WSPR_BENCH_VALU_REUSE=8: cache-resident), per-kernel times by rocprofv3.
set -u
O=gpurun_out/${1:-r06_freq_mem}
mkdir -p $O; rm -f $O/kernel_stats.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in scalar bcast; do
  for reuse in 2048 8; do
    echo "== WSPR_K4_FREQ=$v WSPR_BENCH_VALU_REUSE=$reuse" >> $O/kernel_stats.txt
    rm -rf $O/raw
    WSPR_K4_FREQ=$v WSPR_BENCH_VALU_REUSE=$reuse rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python tools/valu_probe.py 2048 1 5 > $O/probe.log 2> $O/probe.err
    cat $O/probe.log >> $O/kernel_stats.txt
    python tools/profile_summary.py $(ls $O/raw/*/*kernel_trace.csv) | grep -v "at::\|elementwise\|calib" | grep "freq_\|phasor_freq\|demod_lagsys\|sub_\|kernel,grid" >> $O/kernel_stats.txt
    rm -rf $O/raw
  done
done
cat $O/kernel_stats.txt
