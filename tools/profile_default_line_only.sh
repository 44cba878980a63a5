#!/bin/bash
# the driver's own command, unprofiled and timed, + the isolated launch sets' kernel times (a cheap refresh of those two
# profile files after a late kernel change); usage (GPU box): tools/profile_default_line_only.sh r04
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=${1:-r04}; O=gpurun_out/prof_$R; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default_bench_line.json 2> $O/default.err ) 2> $O/default_wall.txt
tools/kprobe.sh $O/isolated 2048 1 > $O/isolated_kernels.txt 2>&1
tail -1 $O/default_bench_line.json | cut -c1-200; cat $O/default_wall.txt; head -6 $O/isolated_kernels.txt
