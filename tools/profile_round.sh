#!/bin/bash
# Round profile set on the GPU box: kernel-trace stats of the three single-GPU configurations, the HBM
# PMC passes of the FFT+sync stage and one SQ pass of the headline configuration.  Everything lands in
# gpurun_out/prof_r02/ ; copy the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --min-seconds 0 > $O/c3_bench.json 2> $O/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python bench.py --config 2 --steps 100 --no-cpu-baseline --min-seconds 0 > $O/c2_bench.json 2> $O/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py --config 5 --segments 128 --steps 4 --warmup 2 --min-seconds 0 > $O/c5_bench.json 2> $O/c5.err
for cfg in "8192 10" "1024 1"; do set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$1 -- python tools/pmc_k1.py $1 $2 > $O/fetch_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$1 -- python tools/pmc_k1.py $1 $2 > $O/write_$1.log 2>&1
  python tools/pmc_summarise.py $(ls $O/fetch_$1/*/*counter_collection.csv) $(ls $O/write_$1/*/*counter_collection.csv) $1 > $O/k1_pmc_traffic_$1.json
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_c3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --min-seconds 0 --inflight 1 > $O/sq_c3_bench.json 2> $O/sq_c3.err
python tools/sq_summarise.py $(ls $O/sq_c3/*/*counter_collection.csv) > $O/sq_c3_summary.json
for c in c3 c2 c5; do python tools/profile_summary.py $(ls $O/$c/*/*kernel_trace.csv) > $O/${c}_kernel_by_grid.csv; cp $(ls $O/$c/*/*kernel_stats.csv) $O/${c}_kernel_stats.csv; done
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
ls -la $O | head -40
