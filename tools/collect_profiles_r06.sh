#!/bin/bash
# gpurun_out/prof_r06/ (tools/profile_r06.sh) -> profiles/r05_*
O=gpurun_out/prof_r06; P=profiles; R=r06
cp $O/default_bench_line.json $P/${R}_default_bench_line.json
cp $O/default_wall.txt $P/${R}_default_bench_wall_time.txt
for c in 3 4; do
  cp $O/c${c}_kernel_stats.csv $P/${R}_config${c}_kernel_stats.csv
  cp $O/c${c}_kernel_by_grid.csv $P/${R}_config${c}_kernel_by_grid.csv
  cp $O/c${c}_gpu_busy.txt $P/${R}_config${c}_gpu_busy.txt
  grep '^{' $O/c${c}_bench.json | tail -1 > $P/${R}_config${c}_bench_under_rocprof.json
done
cp $O/sq_c3_summary.json $P/${R}_config3_sq_counters.json
cp $O/k1_pmc_traffic_8192.json $P/${R}_k1_pmc_traffic.json
cp $O/isolated_kernels.txt $P/${R}_isolated_launch_sets_kernel_times.txt
ls -la $P | grep ${R}_
