#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_scale.py > $OUT/s9_tests.log 2>&1
echo "tests rc=$?" > $OUT/s9_summary.txt
grep -E "passed|failed|^FAILED|^E  " $OUT/s9_tests.log | head -40 >> $OUT/s9_summary.txt
cat $OUT/s9_summary.txt
