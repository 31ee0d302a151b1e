#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "find_copies or fine_stage or smoke or genome_mask or scripts" > $OUT/s2_tests_small.log 2>&1
echo "small tests rc=$?" > $OUT/s2_summary.txt
tail -5 $OUT/s2_tests_small.log >> $OUT/s2_summary.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s > $OUT/s2_tests_scale.log 2>&1
echo "scale tests rc=$?" >> $OUT/s2_summary.txt
grep -E "copy finder|fine stage|coarse stage|C3:|passed|failed|Error|assert" $OUT/s2_tests_scale.log | tail -20 >> $OUT/s2_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 24"
timeout 300 $B > $OUT/s2_bench_default.json 2> $OUT/s2_bench_default.err
python - <<'PY' >> $OUT/s2_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s2_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items()})
        print('   ', d['config']['align_stats_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s2_summary.txt
