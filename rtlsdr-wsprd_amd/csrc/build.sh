#!/bin/bash
# Builds libwspr_mi355x.so IN-TREE for gfx950 (hipcc cross-compiles without a GPU).
# -ffp-contract=off: the decoder's decisions are argmax/threshold tests on float32
# sums evaluated in the reference's order (x86-64 SSE, no FMA); contraction would
# change them.  sqrt/div stay correctly rounded (hipcc default).
set -euo pipefail
trap "echo BUILD FAILED >&2" ERR
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libwspr_mi355x.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result ${WSPR_EXTRA_FLAGS:-}"
mkdir -p "$here/obj"
srcs=(kernels/k0_decimate.hip kernels/k1_fft_bank.hip kernels/k2_k3_sync.hip kernels/k4_demod.hip
      kernels/k6_fano_wave.hip kernels/k7_subtract.hip host/wspr_pipeline.hip host/wspr_capi.hip)
objs=()
pids=()
for s in "${srcs[@]}"; do
  o="$here/obj/$(basename "${s%.*}").o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$here/$s" -nt "$o" ] || [ -n "$(find "$here" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ] \
     || [ "$here/../../include/wspr_mi355x.h" -nt "$o" ]; then
    $HIPCC $FLAGS -x hip -c "$here/$s" -o "$o" &
    pids+=($!)
  fi
done
o="$here/obj/wspr_message.o"
objs+=("$o")
if [ ! -f "$o" ] || [ "$here/host/wspr_message.cpp" -nt "$o" ] || [ -n "$(find "$here/host" -name '*.h' -newer "$o" | head -1)" ]; then
  g++ -O3 -mpopcnt -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-format-truncation -c "$here/host/wspr_message.cpp" -o "$o" &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# exported symbols = exactly the functions and data include/wspr_mi355x.h declares
map="$here/obj/exports.map"
python3 - "$here/../../include/wspr_mi355x.h" > "$map" <<'PY'
import re, sys
src = re.sub(r"/\*.*?\*/", "", open(sys.argv[1]).read(), flags=re.S)
names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)) - {"defined"}
names |= set(re.findall(r"extern\s+[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])+\s*;", src))
print("{ global: " + " ".join(n + ";" for n in sorted(names)) + " local: *; };")
PY
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$map" -o "$out" "${objs[@]}" -lpthread
echo "built $out"
