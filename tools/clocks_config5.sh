cd "$GRAFT_REPO_ROOT"
(for i in $(seq 1 70); do echo "s$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|Package Power" | sed "s/GPU\[0\]\s*: //" | tr "\n" " ")"; sleep 0.4; done > /tmp/clk5.txt) &
python bench.py --config 5 --no-cpu-baseline --no-pmc --min-seconds 6 2>/dev/null | python tools/c5_line.py config5
wait
python - <<'PY'
import re
for ln in open('/tmp/clk5.txt'):
    m=re.search(r'sclk.*?\((\d+)Mhz\)',ln); p=re.search(r'\(W\): ([0-9.]+)',ln); mm=re.search(r'mclk.*?\((\d+)Mhz\)',ln)
    if m and p: print(ln.split()[0], "sclk", m.group(1), "mclk", mm.group(1) if mm else "?", "W", p.group(1))
PY
