# configs[4] (bench.py --config 5) with K0's WAVES at a higher issue priority (s_setprio) than the decoder's, as ordinary
# and as resident grid (lab library; the stream-priority form of the experiment: profiles/r05_k0_priority_ab.txt)
slim="--no-cpu-baseline --no-pmc --no-kernel-roofline"
for rep in 1 2; do
for v in "" "WSPR_K0_WAVEPRIO=3" "WSPR_K0_RESIDENT=1 WSPR_K0_WAVEPRIO=3" "WSPR_K0_RESIDENT=2 WSPR_K0_WAVEPRIO=3" "WSPR_K0_RESIDENT=4 WSPR_K0_WAVEPRIO=3" "WSPR_K0_RESIDENT=2 WSPR_K0_WAVEPRIO=1"; do
  env WSPR_USE_LAB=1 $v timeout 300 python bench.py --config 5 --steps 6 --warmup 2 $slim 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value']), round(d['ms_per_step'],1), d['decoded_ok'])"
done; done
