#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "find_copies or smoke or cons_v1 or deredundant or trmask or pan_remove" > $OUT/s6_tests.log 2>&1
echo "tests rc=$?" > $OUT/s6_summary.txt
tail -12 $OUT/s6_tests.log >> $OUT/s6_summary.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s -k "c2" > $OUT/s6_tests_scale.log 2>&1
echo "scale tests rc=$?" >> $OUT/s6_summary.txt
grep -E "copy finder|fine stage|coarse stage|C3:|passed|failed|Error|assert" $OUT/s6_tests_scale.log | tail -20 >> $OUT/s6_summary.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-coarse --verify 24"
timeout 300 $B > $OUT/s6_bench_default.json 2> $OUT/s6_bench_default.err
cp hite_amd/libhite_gpu.so /tmp/normal.so
cp hite_amd/libhite_gpu_clk.so hite_amd/libhite_gpu.so
timeout 300 $B > $OUT/s6_bench_clk.json 2> $OUT/s6_bench_clk.err
HITE_JUDGE_WAVE_COLS=0 timeout 300 $B > $OUT/s6_bench_clk_block.json 2> /dev/null
cp /tmp/normal.so hite_amd/libhite_gpu.so
python - <<'PY' >> $OUT/s6_summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/s6_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels']
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify',{}).get('mismatches'), 'is_te', d['config']['is_te'], 'copies', d['config']['copies'])
        print('   ', {n:v['ms_per_step'] for n,v in k.items()})
        print('   ', d['config']['copy_stats'], d.get('judge_phase_ticks'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s6_summary.txt
