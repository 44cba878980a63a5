#!/bin/bash
# (GPU box) A/B of the frequency scan: table through the scalar cache (freq_scalar_kernel, default) against the table in
# registers handed out by the matrix pipe (freq_bcast_kernel, WSPR_K4_FREQ=bcast; lab library).  Trace parity of the
# alternative first, then the launch-set times (HIP events) and the per-kernel times (rocprofv3 --kernel-trace --stats).
set -u
O=gpurun_out/${1:-r06_freq_bcast}
mkdir -p $O
export TMPDIR=/tmp
( WSPR_K4_FREQ=bcast WSPR_TRACE_SCENES=60 python tests/trace_parity.py parity scenes config3 2>&1 | tail -5 ) > $O/trace_parity_bcast.txt
for v in scalar bcast; do
  for rep in 1 2 3; do
    echo "== WSPR_K4_FREQ=$v run $rep" >> $O/launch_sets.txt
    WSPR_K4_FREQ=$v python tools/valu_probe.py 2048 1 5 2>/dev/null | tail -1 >> $O/launch_sets.txt
  done
  ( cd /tmp && WSPR_K4_FREQ=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p -- python $GRAFT_REPO_ROOT/tools/valu_probe.py 2048 1 5 > /dev/null 2>&1 )
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== WSPR_K4_FREQ=$v" >> $O/kernel_stats.txt
  python - "$f" >> $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("freq_", "phasor_freq", "demod_lagsys")):
        print("%-60s calls %5s avg %9.2f us total %10.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
cat $O/trace_parity_bcast.txt $O/launch_sets.txt $O/kernel_stats.txt
