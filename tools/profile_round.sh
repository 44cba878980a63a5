#!/bin/bash
# Round profile set on the GPU box (one gpurun call): the driver's own command unprofiled and timed, kernel-trace stats
# of the three single-GPU configurations, the HBM PMC passes of the FFT+sync stage (separate --pmc passes, no other
# trace domain), SQ passes of the headline configuration and of the isolated fp32 launch sets, the K6w latency
# table, the packed-fp32 issue-rate micro-benchmark and the K0 CU-share sweep.  Everything lands in
# gpurun_out/prof_r03/ ; the summaries are copied into profiles/ afterwards (tools/collect_profiles.sh).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=${1:-r03}; O=gpurun_out/prof_$R; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default_bench_line.json 2> $O/default.err ) 2> $O/default_wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --min-seconds 0 > $O/c3_bench.json 2> $O/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python bench.py --config 2 --steps 100 --no-cpu-baseline --no-pmc --min-seconds 0 > $O/c2_bench.json 2> $O/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py --config 5 --segments 512 --steps 3 --warmup 1 --no-cpu-baseline --min-seconds 0 > $O/c5_bench.json 2> $O/c5.err
for cfg in "8192 10" "1024 1"; do set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$1 -- python tools/pmc_k1.py $1 $2 > $O/fetch_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$1 -- python tools/pmc_k1.py $1 $2 > $O/write_$1.log 2>&1
  python tools/pmc_summarise.py $(ls $O/fetch_$1/*/*counter_collection.csv) $(ls $O/write_$1/*/*counter_collection.csv) $1 > $O/k1_pmc_traffic_$1.json
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_c3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --min-seconds 0 --inflight 1 > $O/sq_c3_bench.json 2> $O/sq_c3.err
python tools/sq_summarise.py $(ls $O/sq_c3/*/*counter_collection.csv) > $O/sq_c3_summary.json
for c in c3 c2 c5; do python tools/profile_summary.py $(ls $O/$c/*/*kernel_trace.csv) > $O/${c}_kernel_by_grid.csv; cp $(ls $O/$c/*/*kernel_stats.csv) $O/${c}_kernel_stats.csv; done
python tools/gpu_busy.py $(ls $O/c3/*/*kernel_trace.csv) 0.45 0.9 > $O/c3_gpu_busy.txt 2>&1
# isolated launch sets: per-kernel durations and SQ counters (2 048 single-signal candidates)
tools/kprobe.sh $O/isolated 2048 1 > $O/isolated_kernels.txt 2>&1
tools/sqprobe.sh $O/isolated_sq 2048 1 > $O/isolated_sq.txt 2>&1
bash tools/k1_sq.sh > $O/k1_sq.txt 2>&1
python tools/fano_latency.py > $O/k6w_latency.txt 2>&1
[ -x tools/valu_issue_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value tools/valu_issue_probe.hip -o tools/valu_issue_probe.bin
tools/valu_issue_probe.bin > $O/valu_issue_probe.txt 2>&1
tools/k0_cu_share.sh $O/k0_cus 2 > $O/k0_cu_share.txt 2>&1
python tools/k0_scan.py 32 > $O/k0_alone.txt 2>&1
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete
find $O -name "*.db" -delete; du -sh $O; ls $O | head -60
