F="--no-cpu-baseline --no-coarse --no-modes --verify 0 --steps 6 --warmup 2"
for SC in 1,1,1,1 2,1,1,1 4,1,1,1 1,2,1,1 1,4,1,1 1,1,2,1 1,1,0.5,1 1,1,4,1 0.5,0.5,1,1; do
  HITE_ALIGN_LANES_SCALE=$SC python bench.py --config C4share $F > gpurun_out/s_$SC.json 2>/dev/null
  python - "$SC" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/s_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], {k.replace('align_','').replace('_long',''):v['ms_per_step'] for k,v in d['kernels'].items() if 'align' in k and 'prep' not in k and 'fallback' not in k})
PY
done
