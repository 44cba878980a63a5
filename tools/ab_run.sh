#!/bin/bash
# on the GPU box: tools/ab_run.sh <tag|default> ... : valu_probe + slim bench line per library variant
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
cp rtlsdr-wsprd_amd/libwspr_mi355x.so /tmp/default.so
for tag in "$@"; do
  if [ "$tag" = default ]; then cp /tmp/default.so rtlsdr-wsprd_amd/libwspr_mi355x.so; else cp rtlsdr-wsprd_amd/libwspr_mi355x.so.$tag rtlsdr-wsprd_amd/libwspr_mi355x.so; fi
  echo "== $tag"
  python tools/valu_probe.py 2048 1 5 2>/dev/null
  python bench.py --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings 2>/dev/null | python tools/slim_line.py | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']))"
done
cp /tmp/default.so rtlsdr-wsprd_amd/libwspr_mi355x.so
