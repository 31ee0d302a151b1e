#!/bin/bash
# TEST INFRASTRUCTURE.  Which lines of the CPU oracle do the golden-vector tests execute?  Builds oracle/*.c with gcov
# instrumentation into a scratch directory, runs tests/test_oracle_golden.py (oracle == the reference's outputs) against that
# build, and prints the line coverage per file plus every line never executed: a line of the restatement that no golden
# reaches is a line whose agreement with the reference nobody has checked.
#   bash tools/oracle_line_coverage.sh > profiles/rNN_oracle_line_coverage.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
FILES="hite_oracle hite_oracle_coarse hite_oracle_ltr hite_oracle_lib hite_oracle_itr hite_oracle_msa hite_oracle_nw hite_oracle_copies hite_oracle_trf"
SRC=""; for f in $FILES; do SRC="$SRC $ROOT/oracle/$f.c"; done
cd "$W"
gcc -O0 --coverage -fPIC -std=c11 -shared -o "$W/libhite_oracle.so" $SRC -lm
HITE_ORACLE_SO="$W/libhite_oracle.so" python -m pytest "$ROOT/tests/test_oracle_golden.py" -q -x -p no:cacheprovider > "$W/pytest.log" 2>&1 || { cat "$W/pytest.log"; exit 1; }
echo "# gcov line coverage of oracle/*.c under tests/test_oracle_golden.py ($(tail -1 "$W/pytest.log"))"
echo "# restatements of HiTE's own Python (pinned by the reference's outputs): hite_oracle.c (judges, searches, TSD, sparse columns, gather"
echo "# rules), hite_oracle_coarse.c (FMEA, chaining variants), hite_oracle_ltr.c, hite_oracle_lib.c; the other four files are the CPU twins of"
echo "# the build's own stand-ins for third-party tools (checked HIP == twin in the -m gpu tests; only partly touched by the goldens)"
for f in $FILES; do
    gcov -o "$W/libhite_oracle.so-$f.gcda" "$ROOT/oracle/$f.c" > "$W/$f.sum" 2>/dev/null || true
    printf "%-24s %s\n" "$f.c" "$(grep -A1 "File '$ROOT/oracle/$f.c'" "$W/$f.sum" | tail -1)"
done
echo "# lines never executed in the five restatement files (hite_oracle_itr.c: the restatement of the itrsearch binary, pinned by the tool's own output):"
for f in hite_oracle hite_oracle_coarse hite_oracle_ltr hite_oracle_lib hite_oracle_itr; do
    grep -n "#####" "$W/$f.c.gcov" | sed "s/^[0-9]*: *#####: */$f.c:/" || true
done
rm -rf "$W"
