#!/bin/bash
# configs[4] with the front end confined to a share of the CUs: K0's own rate and the pipeline's throughput.
# usage (GPU box): tools/k0_cu_share.sh <outdir>
O=$1; mkdir -p $O
for cus in 0 64 96 128 192; do
  python bench.py --config 5 --segments 512 --steps 6 --warmup 1 --no-cpu-baseline --k0-cus $cus --inflight ${2:-2} > $O/c5_cus$cus.json 2> $O/c5_cus$cus.err
  python - $O/c5_cus$cus.json $cus <<'PY'
import json,sys
t=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not t: print(sys.argv[2], "no json"); sys.exit()
d=json.loads(t[-1]); k=d["roofline"]["front_end_K0"]
print("k0_cus %4s: %6.0f seg/s, %6.1f ms/step, K0 alone %.2f ms per wave = %.3f of peak, decoded %s" % (sys.argv[2], d["value"], d["ms_per_step"], k["avg_launch_ms"], k["frac"], d["decoded_ok"]))
PY
done
