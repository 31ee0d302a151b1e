#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/s11_tests.log 2>&1
echo "tests rc=$?" > $OUT/s11_summary.txt
grep -E "passed|failed|^FAILED|^E  " $OUT/s11_tests.log | head -40 >> $OUT/s11_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/s11_summary.txt 2>&1
cat $OUT/s11_summary.txt
