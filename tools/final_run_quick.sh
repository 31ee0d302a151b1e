export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/final
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/final/smoke.txt
timeout 400 python bench.py 2>gpurun_out/final/bench_default.err | tail -1 > gpurun_out/final/bench_default.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/kt -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/kt.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/final/kt > gpurun_out/final/kernel_stats.txt 2>&1
rm -rf gpurun_out/final/kt
timeout 300 python bench.py --stage coarse --steps 3 --warmup 2 2>gpurun_out/final/bench_coarse.err | tail -1 > gpurun_out/final/bench_coarse.json
cat gpurun_out/final/pytest_gpu.txt gpurun_out/final/smoke.txt
cut -c1-400 gpurun_out/final/bench_default.json
head -12 gpurun_out/final/kernel_stats.txt
cut -c1-300 gpurun_out/final/bench_coarse.json
