#!/bin/bash
# Copies the summaries of gpurun_out/prof_<round>/ (tools/profile_round.sh) into profiles/ under per-round names.
R=${1:-r04}; O=gpurun_out/prof_$R; P=profiles
cp $O/default_bench_line.json $P/${R}_default_bench_line.json
cp $O/default_wall.txt $P/${R}_default_bench_wall_time.txt
for c in 3 2 5; do
  cp $O/c${c}_kernel_stats.csv $P/${R}_config${c}_kernel_stats.csv
  cp $O/c${c}_kernel_by_grid.csv $P/${R}_config${c}_kernel_by_grid.csv
  grep '^{' $O/c${c}_bench.json | tail -1 > $P/${R}_config${c}_bench_under_rocprof.json
done
cp $O/c3_gpu_busy.txt $P/${R}_config3_gpu_busy.txt; cp $O/c2_gpu_busy.txt $P/${R}_config2_gpu_busy.txt
cp $O/sq_c3_summary.json $P/${R}_config3_sq_counters.json
cp $O/k1_pmc_traffic_8192.json $P/${R}_k1_pmc_traffic.json
cp $O/k1_pmc_traffic_1024.json $P/${R}_k1_pmc_traffic_1024seg.json
cp $O/isolated_kernels.txt $P/${R}_isolated_launch_sets_kernel_times.txt
cp $O/isolated_sq.txt $P/${R}_isolated_launch_sets_sq_counters.txt
cp $O/k6w_latency.txt $P/${R}_k6w_latency.txt
cp $O/k1_sq.txt $P/${R}_k1_sq_counters.txt
cp $O/valu_issue_probe.txt $P/${R}_packed_fp32_issue_rate_by_occupancy.txt
cp $O/k0_cu_share.txt $P/${R}_k0_cu_share_sweep.txt
cp $O/k0_alone.txt $P/${R}_k0_alone.txt
for f in insts_by_kernel.txt k0_insts_per_16_bytes.txt k0_alone_process_1.txt k0_in_config5_process.txt k0_in_default_line.txt k0_decoder_overlap_by_residency.txt config5_by_k0_residency.txt kernel_unit_checks.txt; do [ -f $O/$f ] && cp $O/$f $P/${R}_$f; done
ls -la $P | grep ${R}_
