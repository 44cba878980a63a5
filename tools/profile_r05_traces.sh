#!/bin/bash
# The two kernel traces of tools/profile_r05.sh alone (headline and configs[3]'s shard), re-taken on the round's final code
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=r05; O=gpurun_out/prof_$R; mkdir -p $O; rm -rf $O/c3 $O/c4
slim="--no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --no-shard-block --no-host-entry --no-hashtable-block --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 24 --warmup 2 $slim > $O/c3_bench.json 2> $O/c3.err
python tools/profile_summary.py $(ls $O/c3/*/*kernel_trace.csv | head -1) > $O/c3_kernel_by_grid.csv; cp $(ls $O/c3/*/*kernel_stats.csv | head -1) $O/c3_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c3/*/*kernel_trace.csv | head -1) 0.6 0.95 > $O/c3_gpu_busy.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python bench.py --config 4 --steps 120 --warmup 4 $slim > $O/c4_bench.json 2> $O/c4.err
python tools/profile_summary.py $(ls $O/c4/*/*kernel_trace.csv | head -1) > $O/c4_kernel_by_grid.csv; cp $(ls $O/c4/*/*kernel_stats.csv | head -1) $O/c4_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c4/*/*kernel_trace.csv | head -1) 0.6 0.95 > $O/c4_gpu_busy.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*.db" -delete
head -4 $O/c3_gpu_busy.txt; head -4 $O/c4_gpu_busy.txt; grep fft_bank_avg $O/c3_kernel_by_grid.csv | tail -2; grep '^{' $O/c3_bench.json | tail -1 | cut -c1-200
