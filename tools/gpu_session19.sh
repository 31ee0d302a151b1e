#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/s19_tests.log 2>&1
echo "tests rc=$?" > $OUT/s19_summary.txt
grep -E "passed|failed|^FAILED|^E  " $OUT/s19_tests.log | head -40 >> $OUT/s19_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-coarse --verify 24"
timeout 300 $B > $OUT/s19_bench_c3.json 2> $OUT/s19_bench_c3.err
python - <<'PY' >> $OUT/s19_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s19_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'], 'sum kernels', round(sum(v['ms_per_step'] for v in k.values()),2))
        print('   ', d['config']['align_stats_per_step'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items() if n.startswith('align')})
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s19_summary.txt
