#!/bin/bash
# (GPU box) Is the frequency scan bound by its SAMPLE reads?  The same launch set over (a) 2 048 segments x 1 candidate
# (every candidate its own 332 KB of samples: 680 MB per launch from HBM) and (b) 205 segments x 10 candidates (74 MB:
# the samples stay in the Infinity Cache / L2), per-kernel times by rocprofv3 --kernel-trace --stats.
set -u
O=gpurun_out/${1:-r06_freq_mem}
mkdir -p $O
export TMPDIR=/tmp
for v in scalar bcast; do
  for shape in "2048 1" "205 10"; do
    tag=$(echo $shape | tr ' ' 'x')
    echo "== WSPR_K4_FREQ=$v segments x signals = $shape" >> $O/kernel_stats.txt
    WSPR_K4_FREQ=$v python tools/valu_probe.py $shape 5 2>/dev/null | tail -1 >> $O/kernel_stats.txt
    rm -rf /tmp/prof_$v_$tag
    ( cd /tmp && WSPR_K4_FREQ=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_${v}_$tag -o p -- python $GRAFT_REPO_ROOT/tools/valu_probe.py $shape 5 > /dev/null 2>&1 )
    f=$(find /tmp/prof_${v}_$tag -name "*kernel_stats*" | head -1)
    python - "$f" >> $O/kernel_stats.txt <<'PY'
import csv, sys
if not sys.argv[1]:
    print("(no kernel_stats file)"); sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("freq_", "phasor_freq", "demod_lagsys")):
        print("%-60s calls %5s avg %9.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
find /tmp/prof_scalar_2048x1 | head -20 >> $O/files.txt
cat $O/kernel_stats.txt
