#!/bin/bash
# One profiling session on the GPU box: the bench line, rocprofv3 kernel statistics and the PMC passes (each in its own run,
# --kernel-trace only), summaries written under gpurun_out/ for copying into profiles/.
# usage: tools/profile_round.sh <tag>     (e.g. r02)
set -u
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-coarse --no-modes --verify 0"
timeout 900 python bench.py > $OUT/${TAG}_bench_fine.json 2> $OUT/${TAG}_bench_fine.err
timeout 600 python bench.py --config C2 > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
timeout 900 python bench.py --config C5 --steps 3 --warmup 1 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
if [ "${SKIP_C5X8:-0}" != "1" ]; then
timeout 1200 python bench.py --config C5 --genomes 8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_c5x8.json 2> $OUT/${TAG}_bench_c5x8.err
fi
timeout 600 python bench.py --config C4share > $OUT/${TAG}_bench_c4share.json 2> $OUT/${TAG}_bench_c4share.err
if [ "${QUICK:-0}" != "1" ]; then      # (QUICK=1: the benches and the rocprofv3 passes only)
timeout 600 python -m pytest tests/test_gpu_scale.py -q -m gpu -s -k "c2 or c3 or c5" > $OUT/${TAG}_scale_tests_raw.txt 2>&1
grep -E "(copy finder|fine stage|coarse stage|C3:|C5 merge|cost sums|anchor matches|^\.*C2,)" $OUT/${TAG}_scale_tests_raw.txt | sed "s/^\.*//" > $OUT/${TAG}_scale_tests.txt
timeout 600 python tools/copy_interval_modes.py >> $OUT/${TAG}_scale_tests.txt 2> $OUT/copy_interval_modes.err
fi
HITE_ALIGN_EXACT=16 timeout 600 python bench.py --no-cpu-baseline --no-coarse > $OUT/${TAG}_bench_fine_cap16.json 2> /dev/null
HITE_ALIGN_EXACT=0 timeout 600 python bench.py --no-cpu-baseline --no-coarse > $OUT/${TAG}_bench_fine_cap0.json 2> /dev/null
rm -rf $OUT/prof_stats $OUT/prof_sq $OUT/prof_fetch $OUT/prof_write
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o run -- $BENCH > $OUT/prof_stats.json 2> $OUT/prof_stats.log
timeout 900 rocprofv3 --kernel-trace --kernel-include-regex "align_|judge|star_|chain_" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/prof_sq -o run -- $BENCH > $OUT/prof_sq.json 2> $OUT/prof_sq.log
timeout 900 rocprofv3 --kernel-trace --kernel-include-regex "align_|judge|star_|row_gather|rs_|hit_|occ_|cluster|cand_min|chain_" --pmc FETCH_SIZE -d $OUT/prof_fetch -o run -- $BENCH > $OUT/prof_fetch.json 2> $OUT/prof_fetch.log
timeout 900 rocprofv3 --kernel-trace --kernel-include-regex "align_|judge|star_|row_gather|rs_|hit_|occ_|cluster|cand_min|chain_" --pmc WRITE_SIZE -d $OUT/prof_write -o run -- $BENCH > $OUT/prof_write.json 2> $OUT/prof_write.log
python tools/pmc_counters.py $OUT/prof_sq $OUT/prof_sq.json $OUT/${TAG}_sq_counters.json $OUT/${TAG}_sq_counters.txt "rocprofv3 --kernel-trace --kernel-include-regex align_ --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -- $BENCH" 2> $OUT/pmc_counters.err
python tools/pmc_traffic.py $OUT/prof_fetch $OUT/prof_write $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_hbm.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- $BENCH" 2> $OUT/pmc_traffic.err
python - <<PY > $OUT/${TAG}_kernel_stats.txt 2> $OUT/kernel_stats.err
import glob, sqlite3
dbs = sorted(glob.glob("$OUT/prof_stats/**/*.db", recursive=True))
print("# rocprofv3 --kernel-trace --stats -- $BENCH  (MI355X; 5 pipeline steps + set-up in the run); durations in microseconds")
for db in dbs:
    con = sqlite3.connect(db)
    views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    v = [x for x in views if "top_kernels" in x]
    if not v:
        continue
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % v[0])]
    print("# columns: " + ", ".join(cols))
    for row in con.execute("select * from %s limit 45" % v[0]):
        print("  ".join(str(x)[:70] for x in row))
PY
# the raw rocprofv3 databases stay on the box: gpurun copies back at most 64 MiB
rm -rf $OUT/prof_stats $OUT/prof_sq $OUT/prof_fetch $OUT/prof_write $OUT/cprof_stats $OUT/cprof_fetch $OUT/cprof_write
ls -la $OUT | tail -20
tail -c 600 $OUT/${TAG}_bench_fine.json
