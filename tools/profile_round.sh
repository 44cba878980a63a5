#!/bin/bash
# Round profile set on the GPU box (one gpurun call): the driver's own command unprofiled and timed, kernel-trace stats
# of the three single-GPU configurations, the HBM PMC passes of the FFT+sync stage (separate --pmc passes, no other
# trace domain), SQ passes of the headline configuration and of the isolated fp32 launch sets, the K6w latency
# table, the packed-fp32 issue-rate micro-benchmark and the K0 CU-share sweep.  Everything lands in
# gpurun_out/prof_<round>/ ; the summaries are copied into profiles/ afterwards (tools/collect_profiles.sh).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=${1:-r04}; O=gpurun_out/prof_$R; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default_bench_line.json 2> $O/default.err ) 2> $O/default_wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python bench.py --steps 24 --warmup 2 --no-ceilings --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --min-seconds 0 > $O/c3_bench.json 2> $O/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python bench.py --config 2 --steps 300 --no-cpu-baseline --no-pmc --no-ceilings --min-seconds 0 > $O/c2_bench.json 2> $O/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py --config 5 --segments 512 --steps 3 --warmup 1 --no-cpu-baseline --min-seconds 0 > $O/c5_bench.json 2> $O/c5.err
for cfg in "8192 10" "1024 1"; do set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$1 -- python tools/pmc_k1.py $1 $2 > $O/fetch_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$1 -- python tools/pmc_k1.py $1 $2 > $O/write_$1.log 2>&1
  python tools/pmc_summarise.py $(ls $O/fetch_$1/*/*counter_collection.csv) $(ls $O/write_$1/*/*counter_collection.csv) $1 > $O/k1_pmc_traffic_$1.json
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq_c3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --min-seconds 0 --inflight 1 > $O/sq_c3_bench.json 2> $O/sq_c3.err
python tools/sq_summarise.py $(ls $O/sq_c3/*/*counter_collection.csv) > $O/sq_c3_summary.json
for c in c3 c2 c5; do python tools/profile_summary.py $(ls $O/$c/*/*kernel_trace.csv) > $O/${c}_kernel_by_grid.csv; cp $(ls $O/$c/*/*kernel_stats.csv) $O/${c}_kernel_stats.csv; done
python tools/gpu_busy.py $(ls $O/c3/*/*kernel_trace.csv) 0.6 0.95 > $O/c3_gpu_busy.txt 2>&1
python tools/gpu_busy.py $(ls $O/c2/*/*kernel_trace.csv) 0.45 0.9 > $O/c2_gpu_busy.txt 2>&1
# isolated launch sets: per-kernel durations and SQ counters (2 048 single-signal candidates)
tools/kprobe.sh $O/isolated 2048 1 > $O/isolated_kernels.txt 2>&1
tools/sqprobe.sh $O/isolated_sq 2048 1 > $O/isolated_sq.txt 2>&1
bash tools/k1_sq.sh > $O/k1_sq.txt 2>&1
python tools/fano_latency.py > $O/k6w_latency.txt 2>&1
[ -x tools/valu_issue_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value tools/valu_issue_probe.hip -o tools/valu_issue_probe.bin
tools/valu_issue_probe.bin > $O/valu_issue_probe.txt 2>&1
tools/k0_cu_share.sh $O/k0_cus 2 > $O/k0_cu_share.txt 2>&1
python tools/k0_scan.py 32 > $O/k0_alone.txt 2>&1
# round 4: where the vector issue slots of a configs[2] step go; K0's vector instructions per 16 bytes (both block-sum
# kernels); K0 in three fresh processes; K0 as a resident grid beside the decoder; kernel-level checks with timings
tools/insts_by_kernel.sh $O/insts > $O/insts_by_kernel.log 2>&1; cp $O/insts/insts_by_kernel.txt $O/insts_by_kernel.txt
for kern in mfma dot4; do
  WSPR_K0_KERNEL=$kern rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/k0i_$kern -- python tools/k0_scan.py 8 > $O/k0i_$kern.log 2>&1
  python - $(ls $O/k0i_$kern/*/*counter_collection.csv) $kern >> $O/k0_insts_per_16_bytes.txt <<'PY'
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "SQ_INSTS_VALU": continue
    k = r["Kernel_Name"].split("wspr::(anonymous namespace)::")[-1].split("(")[0]
    acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc:
    if k.startswith("cic_block_sums"):
        per_launch = acc[k] / n[k]
        print("WSPR_K0_KERNEL=%s %s: %.4g vector instructions per launch of 8 segments = %.1f per 16 bytes and lane (%d launches)" %
              (sys.argv[2], k, per_launch, per_launch * 64 / (8 * 576e6 / 16), n[k]))
PY
done
python tools/k0_scan.py 32 > $O/k0_alone_process_1.txt 2>&1
python bench.py --config 5 --no-cpu-baseline --no-pmc 2> /dev/null | python tools/c5_line.py "bench.py --config 5:" > $O/k0_in_config5_process.txt 2>&1
python - $O/default_bench_line.json > $O/k0_in_default_line.txt <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
k = d["tertiary"]["front_end_K0"]
print("default bench line, tertiary block: K0 %.3f ms per %d segments = %.3f of HBM peak; sets of ten: %s; configs[4] %.0f segments/s" %
      (k["avg_launch_ms"], k["segments_per_launch"], k["frac"], k["avg_launch_ms_of_each_set_of_10"], d["tertiary"]["value"]))
PY
for p in 0 1 2 4; do echo "WSPR_K0_RESIDENT=$p"; WSPR_K0_RESIDENT=$p python tools/overlap_probe.py 2>/dev/null | tail -2; done > $O/k0_decoder_overlap_by_residency.txt 2>&1
for p in 0 2 4; do WSPR_K0_RESIDENT=$p python bench.py --config 5 --no-cpu-baseline --no-pmc 2>/dev/null | python tools/c5_line.py "WSPR_K0_RESIDENT=$p:"; done > $O/config5_by_k0_residency.txt 2>&1
( tools/lagsys_check.bin | tail -6; tools/mfma_i8_probe.bin | tail -3; tools/dpp_probe.bin | head -1 ) > $O/kernel_unit_checks.txt 2>&1
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete
find $O -name "*.db" -delete; du -sh $O; ls $O | head -60
