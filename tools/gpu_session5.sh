#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_scale.py > $OUT/s5_tests_small.log 2>&1
echo "small tests rc=$?" > $OUT/s5_summary.txt
tail -15 $OUT/s5_tests_small.log >> $OUT/s5_summary.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s > $OUT/s5_tests_scale.log 2>&1
echo "scale tests rc=$?" >> $OUT/s5_summary.txt
grep -E "copy finder|fine stage|coarse stage|C3:|passed|failed|Error|assert" $OUT/s5_tests_scale.log | tail -20 >> $OUT/s5_summary.txt
timeout 900 python bench.py > $OUT/s5_bench_default.json 2> $OUT/s5_bench_default.err
echo "bench rc=$?" >> $OUT/s5_summary.txt
timeout 600 python bench.py --config C5 --steps 2 --warmup 0 > $OUT/s5_bench_c5.json 2> $OUT/s5_bench_c5.err
echo "c5 rc=$?" >> $OUT/s5_summary.txt
timeout 300 python - > $OUT/s5_trmask_time.txt 2>&1 <<'PY'
import time, torch, numpy as np, hite_amd
from hite_amd import synth
w = synth.make_workload(genome_bp=100_000_000, n_tir=500, n_ltr=0, cands_per_family=1, seed=5, device=torch.device("cuda", 0))
ctx = hite_amd.Context(0)
ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"])
t = time.time(); m = ctx.tr_mask(500); dt = time.time() - t
print("tr_mask 100 Mbp: %.2f s, %d bases masked" % (dt, int(m.sum())))
PY
cat $OUT/s5_trmask_time.txt >> $OUT/s5_summary.txt
python - <<'PY' >> $OUT/s5_summary.txt
import json
for f in ('gpurun_out/s5_bench_default.json','gpurun_out/s5_bench_c5.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], 'verify', d.get('verify'), '\n  coarse', d.get('coarse'), '\n  cpu', d.get('cpu_baseline'), '\n  roofline.step', (d.get('roofline') or {}).get('step'))
        if 'C5' in d['config']['workload']: print('  ', d['config'])
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/s5_summary.txt
