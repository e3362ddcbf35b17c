#!/usr/bin/env bash
# CPU sanitizers over the host-side native code (the GPU pool offers none; this runs in the dev container or on any CPU):
#   1. ThreadSanitizer -- tools/sanitize_driver.cpp = csrc/grid_host.cpp + csrc/pack_host.cpp compiled into one program with
#      clang's OpenMP runtime and its Archer tool (TSAN understands libomp's barriers through it; under libgomp every
#      barrier is a false positive): hexahedra + Kuhn tetrahedra, 1 thread vs a team of 8, all 26 arrays compared.
#   2. AddressSanitizer + UBSan -- the same driver under g++/libgomp, the product's own compiler.
#   3. AddressSanitizer + UBSan -- the CPU test files that exercise the host library and the oracle through their real
#      entry points (ctypes): libninpol_amd.so relinked with instrumented grid_host.o / pack_host.o, the oracle's C file
#      rebuilt instrumented; tests/test_host.py, tests/test_oracle.py, tests/test_grid_invariants.py under LD_PRELOAD=libasan.
# Usage: bash tools/sanitize_host.sh [edge]      (exit code 0 = all three clean); log: profiles/r03/sanitizers.txt
set -u -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
N="${1:-40}"
W=/tmp/nin_sanitize
mkdir -p "$W"
LLVM=/opt/rocm/lib/llvm
fail=0

echo "== 1. ThreadSanitizer (clang + libomp + Archer), edge $N, team 8"
"$LLVM/bin/clang++" -O1 -g -fopenmp -fsanitize=thread -ffp-contract=off -std=c++17 -I /opt/rocm/include \
    "$ROOT/tools/sanitize_driver.cpp" -o "$W/driver_tsan" || fail=1
OMP_TOOL_LIBRARIES="$LLVM/lib/libarcher.so" TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0 exitcode=66" \
    LD_LIBRARY_PATH="$LLVM/lib" "$W/driver_tsan" "$N" 8 2>&1 | tail -n 40 || fail=1

echo "== 2. AddressSanitizer + UBSan (g++ + libgomp), edge $N, team 8"
g++ -O1 -g -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off -std=c++17 \
    -I /opt/rocm/include "$ROOT/tools/sanitize_driver.cpp" -o "$W/driver_asan" || fail=1
ASAN_OPTIONS=detect_leaks=1 "$W/driver_asan" "$N" 8 2>&1 | tail -n 40 || fail=1

echo "== 3. AddressSanitizer + UBSan through ctypes: tests/test_host.py, test_oracle.py, test_grid_invariants.py"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
( cd "$ROOT" && python -c "import __graft_entry__ as g; g.build()" >/dev/null ) || fail=1
for f in grid_host pack_host; do
    g++ $SAN -fopenmp -ffp-contract=off -fPIC -std=c++17 -I /opt/rocm/include -c "$ROOT/ninpol_amd/csrc/$f.cpp" -o "$W/$f.o" || fail=1
done
OBJ="$ROOT/ninpol_amd/csrc/_obj"
others=$(ls "$OBJ"/*.o | grep -v -e grid_host.o -e pack_host.o)
g++ -shared -o "$W/libninpol_amd.so" "$W/grid_host.o" "$W/pack_host.o" $others -L /opt/rocm/lib -lamdhip64 -lgomp \
    -Wl,-rpath,/opt/rocm/lib || fail=1
gcc $SAN -ffp-contract=off -fopenmp -shared -fPIC -std=c99 -o "$W/libninpol_oracle.so" "$ROOT/oracle/ninpol_oracle.c" -lm || fail=1
( cd "$ROOT" && LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 \
    NINPOL_AMD_LIB="$W/libninpol_amd.so" NINPOL_ORACLE_LIB="$W/libninpol_oracle.so" \
    python -m pytest tests/test_host.py tests/test_oracle.py tests/test_grid_invariants.py -q -x -m "not gpu" -p no:cacheprovider 2>&1 | tail -n 25 ) || fail=1
echo "== sanitizers: $([ $fail = 0 ] && echo CLEAN || echo FINDINGS)"
exit $fail
