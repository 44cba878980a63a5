cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for cfg in 3 2; do for s in 3 4 6 8; do
WSPR_SLOTS=$s python bench.py --config $cfg --slots $s --inflight 1 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config $cfg WSPR_SLOTS=$s, one call in flight: ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']), 'ok', d.get('decoded_ok'), 'slots_used', d['config'].get('slots_per_batch'))"
done; done; done
