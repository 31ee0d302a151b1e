#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_scale.py > $OUT/s12_tests.log 2>&1
echo "tests rc=$?" > $OUT/s12_summary.txt
grep -E "passed|failed|^FAILED|^E  " $OUT/s12_tests.log | head -40 >> $OUT/s12_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-coarse --verify 24"
timeout 300 $B > $OUT/s12_bench_c3.json 2> $OUT/s12_bench_c3.err
timeout 300 $B --config C2 > $OUT/s12_bench_c2.json 2> /dev/null
python - <<'PY' >> $OUT/s12_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s12_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items() if n.startswith(('judge','chain'))})
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s12_summary.txt
