#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_scale.py > $OUT/s10_tests.log 2>&1
echo "tests rc=$?" > $OUT/s10_summary.txt
tail -12 $OUT/s10_tests.log >> $OUT/s10_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-coarse --verify 48"
timeout 300 $B > $OUT/s10_bench_default.json 2> $OUT/s10_bench_default.err
python - <<'PY' >> $OUT/s10_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s10_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s10_summary.txt
