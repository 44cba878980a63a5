#!/bin/bash
# (GPU box) configs[4] end to end (bench.py --config 5) under the overlap knobs that exist -- K0 as a thin resident grid of one
# workgroup per CU on a high-priority stream (the one setting that overlapped in isolation: 0.83 of the sum, round 4) -- by
# the number of batches in flight.  Round 6's bounded look before closing the item (verdict item 2).
slim="--no-cpu-baseline --no-pmc --no-kernel-roofline --no-warm-extra"
for rep in 1 2; do
for v in "" "WSPR_K0_RESIDENT=1 WSPR_K0_PRIO=1"; do
for fl in 2 3 4 6 8; do
  env WSPR_USE_LAB=1 $v timeout 300 python bench.py --config 5 --steps 6 --warmup 2 --inflight $fl $slim 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] inflight $fl:', round(d['value']), 'segments/s', round(d['ms_per_step'],1), 'ms', d['decoded_ok'])"
done; done; done
