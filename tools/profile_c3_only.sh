cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=r04; O=gpurun_out/prof_$R; mkdir -p $O; rm -rf $O/c3 $O/sq_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --min-seconds 0 > $O/c3_bench.json 2> $O/c3.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_c3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --min-seconds 0 --inflight 1 > $O/sq_c3_bench.json 2> $O/sq_c3.err
python tools/sq_summarise.py $(ls $O/sq_c3/*/*counter_collection.csv | head -1) > $O/sq_c3_summary.json
python tools/profile_summary.py $(ls $O/c3/*/*kernel_trace.csv | head -1) > $O/c3_kernel_by_grid.csv; cp $(ls $O/c3/*/*kernel_stats.csv | head -1) $O/c3_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c3/*/*kernel_trace.csv | head -1) 0.6 0.95 > $O/c3_gpu_busy.txt 2>&1
ls $O/c3/*/ | head; cat $O/c3_gpu_busy.txt | head -12
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*.db" -delete
