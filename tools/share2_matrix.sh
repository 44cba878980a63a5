slim="--no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --no-shard-block --no-host-entry --no-hashtable-block --no-kernel-roofline --no-warm-extra"
for rep in 1 2; do
for fl in 8 12 16; do
  timeout 300 python bench.py --config 3 --cpu-share 2 --steps 12 --warmup 3 --inflight $fl $slim 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['stage_ms_last_step']; print('cpu-share 2 inflight $fl:', round(d['value']), 'seg/s', round(d['ms_per_step'],1), 'ms; cpu_ms_call', round(t['cpu_ms_call'],1), 'books', round(t['cpu_ms_books'],1), 'refine', round(t['cpu_ms_refine'],1))"
done; done
timeout 300 python bench.py --config 3 --steps 12 --warmup 3 $slim 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['stage_ms_last_step']; print('full host inflight 12:', round(d['value']), 'seg/s', round(d['ms_per_step'],1), 'ms; cpu_ms_call', round(t['cpu_ms_call'],1))"
