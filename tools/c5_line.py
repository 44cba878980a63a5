import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["roofline"].get("front_end_K0", {})
print(sys.argv[1], "value %.0f seg/s, %.1f ms/step, K0 alone frac %.3f (%.2f ms per %d segs)" % (d["value"], d["ms_per_step"], k.get("frac", 0), k.get("avg_launch_ms", 0), k.get("segments_per_launch", 0)))
