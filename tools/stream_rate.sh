# How fast the session path takes receivers' bytes: examples/wspr_host -i on 1 / 4 / 16 / 64 copies of a two-slot raw
# stream (2 x 576 MB each; 65 536-byte callbacks through wspr_session_feed() / wspr_session_feed_many(), all completed
# buffers decoded together at the roll-over)
set -e
make -s -C examples
python - <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
raw, exp = bench.synth_raw_gpu(2, 97531, torch.device("cuda", 0), snr_db=-15.0)
with open("/tmp/two_slots.u8", "wb") as fh:
    for s in range(2):
        fh.write(raw[s].cpu().numpy().tobytes())
print("expected", exp)
PY
for n in 1 4 16 64; do
  args=""; for k in $(seq $n); do args="$args -i /tmp/two_slots.u8"; done
  t0=$(date +%s.%N)
  examples/wspr_host -f 14095600 $args -T 1700000040 | tail -2
  t1=$(date +%s.%N)
  python -c "w=$t1-$t0; print('$n receiver(s): %.2f s wall for 2 slots each (240 s of signal per receiver): %.0f x real time per receiver, %.0f receiver-seconds per second' % (w, 240/w, $n*240/w))"
done
rm -f /tmp/two_slots.u8
