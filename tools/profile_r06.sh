#!/bin/bash
# Round 6's profile set in one gpurun call (GPU box): the driver's command unprofiled and timed; rocprofv3 --kernel-trace
# --stats of the headline configuration and of configs[3]'s per-rank shard (bench.py --config 4) with the GPU-busy
# figures; the HBM PMC passes of the FFT+sync stage (separate --pmc passes); the SQ pass of the headline; the isolated
# launch sets.  Copy the summaries with tools/collect_profiles_r06.sh afterwards.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=r06; O=gpurun_out/prof_$R; mkdir -p $O
slim="--no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --no-shard-block --no-host-entry --no-hashtable-block --no-warm-extra --min-seconds 0"
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default_bench_line.json 2> $O/default.err ) 2> $O/default_wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 24 --warmup 2 $slim > $O/c3_bench.json 2> $O/c3.err
python tools/profile_summary.py $(ls $O/c3/*/*kernel_trace.csv | head -1) > $O/c3_kernel_by_grid.csv; cp $(ls $O/c3/*/*kernel_stats.csv | head -1) $O/c3_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c3/*/*kernel_trace.csv | head -1) 0.6 0.95 > $O/c3_gpu_busy.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python bench.py --config 4 --steps 120 --warmup 4 $slim > $O/c4_bench.json 2> $O/c4.err
python tools/profile_summary.py $(ls $O/c4/*/*kernel_trace.csv | head -1) > $O/c4_kernel_by_grid.csv; cp $(ls $O/c4/*/*kernel_stats.csv | head -1) $O/c4_kernel_stats.csv
python tools/gpu_busy.py $(ls $O/c4/*/*kernel_trace.csv | head -1) 0.6 0.95 > $O/c4_gpu_busy.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python tools/pmc_k1.py 8192 10 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python tools/pmc_k1.py 8192 10 > $O/write.log 2>&1
python tools/pmc_summarise.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) 8192 > $O/k1_pmc_traffic_8192.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_c3 -- python bench.py --steps 2 --warmup 1 $slim --inflight 1 > $O/sq_c3_bench.json 2> $O/sq_c3.err
python tools/sq_summarise.py $(ls $O/sq_c3/*/*counter_collection.csv | head -1) > $O/sq_c3_summary.json
tools/kprobe.sh $O/isolated 2048 1 > $O/isolated_kernels.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*.db" -delete
tail -1 $O/default_bench_line.json | cut -c1-160; cat $O/default_wall.txt; head -5 $O/c3_gpu_busy.txt; head -5 $O/c4_gpu_busy.txt; head -4 $O/isolated_kernels.txt; cat $O/k1_pmc_traffic_8192.json | cut -c1-300
