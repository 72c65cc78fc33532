#!/bin/bash
# Host-side sanitizer run (CPU box).  Builds libpangnn_hip_san.so / libpangnn_torch_san.so (csrc/Makefile, `san`:
# the host halves of the .hip files and torch_ops.cpp under AddressSanitizer + UBSan; device code uninstrumented) and runs the
# CPU tests that drive the C ABI and the dispatcher registration through them.  The python binary is not instrumented, so ROCm
# clang's shared ASan runtime is preloaded; leak checking is off (the interpreter and torch never free their arenas).
# usage: bash tools/run_sanitized.sh [log]     (default log: profiles/r05_host_sanitizer_run.txt)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r05_host_sanitizer_run.txt}
make -C "$ROOT/pangnn_amd/csrc" -j6 san > /tmp/pangnn_san_build.log 2>&1 || { tail -30 /tmp/pangnn_san_build.log; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
{
  echo "# host sanitizer run: $(date -u +%Y-%m-%dT%H:%MZ), $(git -C "$ROOT" rev-parse --short HEAD 2>/dev/null)"
  echo "# libraries: pangnn_amd/libpangnn_hip_san.so, libpangnn_torch_san.so (-fsanitize=address,undefined -fno-sanitize-recover=undefined, host code only)"
  echo "# runtime: $RT (LD_PRELOAD), ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_odr_violation=0  UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1"
} > "$LOG"
cd "$ROOT"
SANENV=(env LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0:detect_odr_violation=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
        PANGNN_HIP_LIB="$ROOT/pangnn_amd/libpangnn_hip_san.so" PANGNN_TORCH_LIB="$ROOT/pangnn_amd/libpangnn_torch_san.so")
"${SANENV[@]}" python - >> "$LOG" 2>&1 <<'PY'
import ctypes
from pangnn_amd import _lib, torch_ops
print("# loaded:", _lib.LIB_PATH, "|", torch_ops.TLIB_PATH)
print("# ASan runtime live in this process:", hasattr(ctypes.CDLL(None), "__asan_init"))
PY
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0:detect_odr_violation=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  PANGNN_HIP_LIB="$ROOT/pangnn_amd/libpangnn_hip_san.so" PANGNN_TORCH_LIB="$ROOT/pangnn_amd/libpangnn_torch_san.so" \
  python -m pytest tests/test_c_abi_host.py tests/test_oracle_cpu.py tests/test_torch_ops.py tests/test_deferred.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tee -a "$LOG"
rc=${PIPESTATUS[0]}
echo "# exit code $rc; sanitizer reports in this log: $(grep -c -E 'ERROR: AddressSanitizer|runtime error:' "$LOG")" >> "$LOG"
exit $rc
