slim="--no-cpu-baseline --no-pmc --no-kernel-roofline"
for rep in 1 2; do
for v in "" "WSPR_K0_PRIO=1" "WSPR_K0_RESIDENT=2" "WSPR_K0_RESIDENT=2 WSPR_K0_PRIO=1" "WSPR_K0_RESIDENT=1 WSPR_K0_PRIO=1" "WSPR_K0_RESIDENT=4 WSPR_K0_PRIO=1"; do
  env WSPR_USE_LAB=1 $v timeout 300 python bench.py --config 5 --steps 6 --warmup 2 $slim 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value']), round(d['ms_per_step'],1), d['decoded_ok'])"
done; done
