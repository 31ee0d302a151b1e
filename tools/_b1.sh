F="--no-cpu-baseline --no-coarse --no-modes --verify 0 --steps 10 --warmup 2"
python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -2
for cfg in ${CFGS:-C4share C2 C3}; do
  for L in ${LS:--2 -1}; do
    HITE_ALIGN_LANES=$L python bench.py --config $cfg $F > gpurun_out/b_${cfg}_L${L}.json 2> gpurun_out/b_${cfg}_L${L}.err
  done
done
HITE_ALIGN_DEBUG=1 python bench.py --config C4share --no-cpu-baseline --no-coarse --no-modes --verify 0 --steps 1 --warmup 0 2>&1 | grep "^\[align" | tail -4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*_L*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'align' in k})
PY
