#!/bin/bash
# configs[3]'s per-rank shard (bench.py --config 4) by batches in flight and CPU share; one JSON line per run
out=${1:-gpurun_out/shard_sweep.jsonl}
: > "$out"
slim="--no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --no-shard-block --no-host-entry --no-hashtable-block --no-kernel-roofline"
for share in 0 2; do
  for inf in 1 2 4 8 12; do
    extra=""; [ "$share" != 0 ] && extra="--cpu-share $share"
    echo "== share $share inflight $inf" >&2
    timeout 300 python bench.py --config 4 --steps 20 --warmup 4 --inflight $inf $extra $slim 2>/dev/null | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'share':$share,'inflight':$inf,'value':d['value'],'ms_per_step':d['ms_per_step'],'decoded_ok':d['decoded_ok'],'stage':d['stage_ms_last_step'],'workers':d['host_pool_workers']}))" >> "$out"
    tail -1 "$out" >&2
  done
done
