cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=r04; O=gpurun_out/prof_$R; mkdir -p $O; rm -rf $O/c2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python bench.py --config 2 --steps 300 --warmup 8 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --min-seconds 0 > $O/c2_bench.json 2> $O/c2.err
python tools/profile_summary.py $(ls $O/c2/*/*kernel_trace.csv | head -1) > $O/c2_kernel_by_grid.csv; cp $(ls $O/c2/*/*kernel_stats.csv | head -1) $O/c2_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c2/*/*kernel_trace.csv | head -1) 0.45 0.9 > $O/c2_gpu_busy.txt 2>&1
cat $O/c2_gpu_busy.txt | head -20; python tools/slim_line.py < $O/c2_bench.json | cut -c1-300
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*.db" -delete
