#!/bin/bash
# Builds the two libraries IN-TREE for gfx950 (hipcc cross-compiles without a GPU):
#   libwspr_mi355x.so      the PRODUCT: exports exactly what include/wspr_mi355x.h declares (the drop-in for wsprd.h)
#   libwspr_mi355x_lab.so  the same sources with -DWSPR_LAB: additionally the stage-level parity hooks, the trace, the
#                          kernel timings and calibration kernels of include/wspr_mi355x_bench.h and the environment
#                          switches that select alternative kernels (tests/ and bench.py's kernel-level measurements)
# -ffp-contract=off: the decoder's decisions are argmax/threshold tests on float32
# sums evaluated in the reference's order (x86-64 SSE, no FMA); contraction would
# change them.  sqrt/div stay correctly rounded (hipcc default).
set -euo pipefail
trap "echo BUILD FAILED >&2" ERR
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
inc="$here/../../include"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result ${WSPR_EXTRA_FLAGS:-}"
srcs=(kernels/k0_decimate.hip kernels/k1_fft_bank.hip kernels/k2_k3_sync.hip kernels/k4_demod.hip
      kernels/k6_fano_wave.hip kernels/k7_subtract.hip host/wspr_context.hip host/wspr_pipeline.hip host/wspr_capi.hip)

# $1 = object directory, $2 = extra compile flags: compiles what is out of date, in parallel; object list in $objs
compile_variant() {
  local objdir="$1" extra="$2" pids=() s o p
  mkdir -p "$objdir"
  objs=()
  for s in "${srcs[@]}"; do
    o="$objdir/$(basename "${s%.*}").o"
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$here/$s" -nt "$o" ] || [ -n "$(find "$here" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ] \
       || [ "$inc/wspr_mi355x.h" -nt "$o" ] || [ "$inc/wspr_mi355x_bench.h" -nt "$o" ] || [ "$here/build.sh" -nt "$o" ]; then
      $HIPCC $FLAGS $extra -x hip -c "$here/$s" -o "$o" &
      pids+=($!)
    fi
  done
  for s in wspr_message wspr_hashmem; do            # pure host C++ (no HIP): the message layer, the batch hash memory
    o="$objdir/$s.o"
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$here/host/$s.cpp" -nt "$o" ] || [ -n "$(find "$here/host" -name '*.h' -newer "$o" | head -1)" ]; then
      g++ -O3 -mpopcnt -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-format-truncation $extra -c "$here/host/$s.cpp" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do if [ -n "$p" ]; then wait "$p"; fi; done   # (nothing out of date: no pids, and no failure)
}

# $1 = map file, $2... = headers: exported symbols = exactly the functions and data those headers declare
export_map() {
  local map="$1"; shift
  python3 - "$@" > "$map" <<'PY'
import re, sys
names = set()
for path in sys.argv[1:]:
    src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    names |= set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)) - {"defined"}
    names |= set(re.findall(r"extern\s+[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])+\s*;", src))
print("{ global: " + " ".join(n + ";" for n in sorted(names)) + " local: *; };")
PY
}

link_variant() {   # $1 = output, $2 = map
  $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$2" -Wl,-Bsymbolic-functions -o "$1" "${objs[@]}" -lpthread
  echo "built $1"
}

which="${1:-all}"
if [ "$which" = all ] || [ "$which" = product ]; then
  compile_variant "$here/obj" ""
  export_map "$here/obj/exports.map" "$inc/wspr_mi355x.h"
  link_variant "$here/../libwspr_mi355x.so" "$here/obj/exports.map"
fi
if [ "$which" = all ] || [ "$which" = lab ]; then
  compile_variant "$here/obj_lab" "-DWSPR_LAB"
  export_map "$here/obj_lab/exports.map" "$inc/wspr_mi355x.h" "$inc/wspr_mi355x_bench.h"
  link_variant "$here/../libwspr_mi355x_lab.so" "$here/obj_lab/exports.map"
fi
