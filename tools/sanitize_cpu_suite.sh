#!/bin/bash
# TEST INFRASTRUCTURE.  The CPU test suite with the oracle (gcc) and the host-compiled device functions (g++,
# tests/test_host_compiled.py) built with AddressSanitizer + UndefinedBehaviorSanitizer; prints the pytest summary and the
# number of sanitizer reports (expected: 0).
#   bash tools/sanitize_cpu_suite.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
SRC=$(ls "$ROOT"/oracle/*.c)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -std=c11 -shared -o "$W/libhite_oracle.so" $SRC -lm
ASAN=$(gcc -print-file-name=libasan.so)
cd "$W"
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 HITE_HOST_CXXFLAGS="-fsanitize=address,undefined -g" \
    HITE_ORACLE_SO="$W/libhite_oracle.so" python -m pytest "$ROOT/tests" -q -m "not gpu" -p no:cacheprovider -s > "$W/out.txt" 2>&1 || true
tail -1 "$W/out.txt"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$W/out.txt" || true)"
rm -rf "$W"
