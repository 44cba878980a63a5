#!/bin/bash
# Shader clock and socket power sampled twice a second while the headline benchmark runs (GPU box).
# usage: tools/clocks_under_load.sh <outfile>
O=${1:-gpurun_out/clocks_under_load.txt}
(for i in $(seq 1 44); do echo "sample $i (0.5 s apart)  $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed "s/GPU\[0\]\s*: //" | tr "\n" " ")"; sleep 0.5; done > $O.samples) &
python bench.py --no-cpu-baseline --no-secondary --no-tertiary --no-pmc --no-share-block --no-reference-case --no-ceilings --min-seconds 8 2>/dev/null | python tools/slim_line.py | cut -c1-90 > $O
wait; cat $O.samples >> $O; rm -f $O.samples; cat $O
