#!/bin/bash
# Sanitizer builds of the LAB library's HOST code + tools/sanitize_driver.cpp (device code is not instrumented).
#   tools/sanitize.sh build asan|tsan     -> build/sanitize/<kind>/{libwspr_mi355x_lab.so, driver}
#   tools/sanitize.sh run asan|tsan host|gpu [nseg]
# ASan + UBSan: heap/stack errors and undefined behaviour on every path the driver walks; TSan: data races between the
# library's own threads (slots, lanes, pools, session).  The HIP runtime is not instrumented: TSan sees its queues
# through their pthread / atomic interfaces only.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cmd="${1:-build}"; kind="${2:-asan}"
out="$root/build/sanitize/$kind"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CXX=/opt/rocm/lib/llvm/bin/clang++
rt="$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")"
case "$kind" in
  asan) SAN="-fsanitize=address,undefined -fno-sanitize=vptr -fno-gpu-sanitize" ;;
  tsan) SAN="-fsanitize=thread -fno-gpu-sanitize" ;;
  *) echo "asan or tsan"; exit 2 ;;
esac
if [ "$cmd" = build ]; then
  mkdir -p "$out"
  FLAGS="--offload-arch=gfx950 -O2 -gline-tables-only -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-omit-frame-pointer -DWSPR_LAB -w"
  src="$root/rtlsdr-wsprd_amd/csrc"
  pids=()
  for s in kernels/k0_decimate.hip kernels/k1_fft_bank.hip kernels/k2_k3_sync.hip kernels/k4_demod.hip kernels/k6_fano_wave.hip \
           kernels/k7_subtract.hip host/wspr_context.hip host/wspr_pipeline.hip host/wspr_capi.hip; do
    $HIPCC $FLAGS $SAN -x hip -c "$src/$s" -o "$out/$(basename "${s%.*}").o" & pids+=($!)
  done
  for s in wspr_message wspr_hashmem; do
    $CXX -O2 -gline-tables-only -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -mpopcnt -w ${SAN/-fno-gpu-sanitize/} \
         -c "$src/host/$s.cpp" -o "$out/$s.o" & pids+=($!)
  done
  for p in "${pids[@]}"; do wait "$p"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC ${SAN/-fno-gpu-sanitize/} -shared-libsan -o "$out/libwspr_mi355x_lab.so" "$out"/*.o -lpthread
  $CXX -O1 -g -std=c++17 -fno-omit-frame-pointer ${SAN/-fno-gpu-sanitize/} -shared-libsan -I/opt/rocm/include "$root/tools/sanitize_driver.cpp" \
       -L"$out" -lwspr_mi355x_lab -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,"$out" -Wl,-rpath,"$rt" -lpthread -o "$out/driver"
  echo "built $out/driver"
else
  mode="${3:-host}"; nseg="${4:-384}"
  export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0" UBSAN_OPTIONS="print_stacktrace=1"
  export TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1:report_signal_unsafe=0"
  log="$(realpath -m "${5:-/tmp/wspr_sanitize_${kind}_${mode}.log}")"
  cd "$out" && LD_LIBRARY_PATH="$out:$rt:/opt/rocm/lib:${LD_LIBRARY_PATH:-}" ./driver "$mode" "$nseg" > "$log" 2>&1 || true
  python3 "$root/tools/sanitize_summary.py" "$log"
fi
