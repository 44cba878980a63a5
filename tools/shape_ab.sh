#!/bin/bash
# on the GPU box: tools/shape_ab.sh <config> "<slots> <inflight>" ... : one bench step line per pipeline shape, two rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
cfg=$1; shift
shapes=("$@")
for round in 1 2; do
for shape in "${shapes[@]}"; do
  set -- $shape
  python bench.py --config $cfg --slots $1 --inflight $2 --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config $cfg slots $1 inflight $2: ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']), 'ok', d.get('decoded_ok'), 'false', d.get('false_decodes'), 'slots_used', d['config'].get('slots_per_batch'))"
done; done
