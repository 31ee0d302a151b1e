F="--no-cpu-baseline --no-coarse --no-modes --verify 0 --steps 10 --warmup 2"
for WC in 2544 2300 2000 1700 1400; do
  HITE_JUDGE_WAVE_COLS=$WC python bench.py --config C3 $F > gpurun_out/w_$WC.json 2>/dev/null
  python - $WC <<'PY'
import json,sys
d=json.loads(open('gpurun_out/w_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("wave cols", sys.argv[1], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'judge' in k})
PY
done
for WR in 48 32; do
  HITE_JUDGE_WAVE_ROWS=$WR python bench.py --config C3 $F > gpurun_out/w_r$WR.json 2>/dev/null
  python - r$WR <<'PY'
import json,sys
d=json.loads(open('gpurun_out/w_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("wave rows", sys.argv[1], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'judge' in k})
PY
done
