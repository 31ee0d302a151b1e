#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "find_copies or smoke" > $OUT/s4_tests.log 2>&1
echo "tests rc=$?" > $OUT/s4_summary.txt
tail -3 $OUT/s4_tests.log >> $OUT/s4_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 24"
timeout 300 $B > $OUT/s4_bench_default.json 2> $OUT/s4_bench_default.err
rm -rf $OUT/s4_prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/s4_prof -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify 0 > $OUT/s4_prof.json 2> $OUT/s4_prof.log
python tools/rocpd_summary.py $OUT/s4_prof > $OUT/s4_kernel_stats.txt 2>&1 || true
python - <<'PY' >> $OUT/s4_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s4_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items()})
        print('   ', d['config']['copy_stats'])
    except Exception as e:
        print(f, 'ERR', e)
PY
head -40 $OUT/s4_kernel_stats.txt >> $OUT/s4_summary.txt
cat $OUT/s4_summary.txt
