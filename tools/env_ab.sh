#!/bin/bash
# on the GPU box: tools/env_ab.sh <config> "VAR=value ..." ... : one bench step line per environment, two rounds ("-" = nothing set)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
cfg=$1; shift
envs=("$@")
for round in 1 2; do
for e in "${envs[@]}"; do
  [ "$e" = "-" ] && e="WSPR_NOTHING=1"
  env $e python bench.py --config $cfg --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); v=d.get('roofline',{}).get('valu',{})
        print('config $cfg [$e]: ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']), 'ok', d.get('decoded_ok'), 'false', d.get('false_decodes'), [round(x['frac_of_no_fma_bound'],3) for x in v.values() if isinstance(x,dict)])"
done; done
