#!/bin/bash
# (GPU box) What does each ingredient of the frequency scan cost?  freq_bcast_kernel with parts removed (WSPR_FQ_ABLATE:
# 1 no re-staging of samples / no barriers after chunk 0, 2 no table loads, 4 no MFMA, 8 no tile reads; sums wrong, timing
# only), per-kernel times by rocprofv3 --kernel-trace.
set -u
O=gpurun_out/${1:-r06_freq_ablate}
mkdir -p $O; rm -f $O/kernel_times.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for ab in 0 1 2 3 7 11 15; do
  rm -rf $O/raw
  WSPR_K4_FREQ=bcast WSPR_FQ_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python tools/valu_probe.py 2048 1 5 > $O/probe.log 2> $O/probe.err
  echo "ablate $ab: $(python tools/profile_summary.py $(ls $O/raw/*/*kernel_trace.csv) | grep freq_bcast)" >> $O/kernel_times.txt
  rm -rf $O/raw
done
cat $O/kernel_times.txt
