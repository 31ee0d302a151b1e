#!/bin/bash
# round-3 GPU session 1: parity tests of the new paths, then bench variants
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "not c3 and not c2" > $OUT/s1_tests_small.log 2>&1
echo "small tests rc=$?" > $OUT/s1_summary.txt
tail -5 $OUT/s1_tests_small.log >> $OUT/s1_summary.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -s > $OUT/s1_tests_scale.log 2>&1
echo "scale tests rc=$?" >> $OUT/s1_summary.txt
tail -12 $OUT/s1_tests_scale.log >> $OUT/s1_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 24"
timeout 300 $B > $OUT/s1_bench_default.json 2> $OUT/s1_bench_default.err
HITE_JUDGE_WAVE_COLS=0 timeout 300 $B > $OUT/s1_bench_blockonly.json 2> /dev/null
HITE_JUDGE_OVERLAP=0 timeout 300 $B > $OUT/s1_bench_nooverlap.json 2> /dev/null
HITE_JUDGE_WAVE_ROWS=32 timeout 300 $B > $OUT/s1_bench_rows32.json 2> /dev/null
HITE_JUDGE_WAVE_COLS=12000 timeout 300 $B > $OUT/s1_bench_cols12000.json 2> /dev/null
python - <<'PY' >> $OUT/s1_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s1_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s1_summary.txt
