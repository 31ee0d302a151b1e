#!/bin/bash
# Stage 3.1 on the GPU box: the coarse bench line (inner + end to end), rocprofv3 kernel statistics and the HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE, each in its own --kernel-trace run), summaries under gpurun_out/ for copying into profiles/.
# usage: tools/profile_coarse.sh <tag>     (e.g. r05)
set -u
TAG=${1:-r05}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --stage coarse --steps 2 --warmup 2 --no-cpu-baseline"
timeout 900 python bench.py --stage coarse --steps 3 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_coarse.json 2> $OUT/${TAG}_bench_coarse.err
rm -rf $OUT/cprof_stats $OUT/cprof_fetch $OUT/cprof_write
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/cprof_stats -o run -- $BENCH > $OUT/cprof_stats.json 2> $OUT/cprof_stats.log
if [ "${2:-}" != "nopmc" ]; then
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cprof_fetch -o run -- $BENCH > $OUT/cprof_fetch.json 2> $OUT/cprof_fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cprof_write -o run -- $BENCH > $OUT/cprof_write.json 2> $OUT/cprof_write.log
python tools/pmc_traffic.py $OUT/cprof_fetch $OUT/cprof_write $OUT/${TAG}_pmc_traffic_coarse.json $OUT/${TAG}_pmc_hbm_coarse.txt "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- $BENCH" 2> $OUT/pmc_traffic_coarse.err
fi
python - <<PY > $OUT/${TAG}_kernel_stats_coarse.txt 2> $OUT/kernel_stats_coarse.err
import glob, sqlite3
dbs = sorted(glob.glob("$OUT/cprof_stats/**/*.db", recursive=True))
print("# rocprofv3 --kernel-trace --stats -- $BENCH  (MI355X, 1 Gbp; in the run: 8 inner steps (index + seeding + FMEA) and 4 end-to-end steps (pack + tandem masking + prev_TE masking incl. a second index build + inner + flanks), warm-up included, + the generation of the genome by torch); durations in microseconds")
for db in dbs:
    con = sqlite3.connect(db)
    views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    v = [x for x in views if "top_kernels" in x]
    if not v:
        continue
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % v[0])]
    print("# columns: " + ", ".join(cols))
    for row in con.execute("select * from %s limit 60" % v[0]):
        print("  ".join(str(x)[:70] for x in row))
PY
# the raw rocprofv3 databases stay on the box: gpurun copies back at most 64 MiB
rm -rf $OUT/prof_stats $OUT/prof_sq $OUT/prof_fetch $OUT/prof_write $OUT/cprof_stats $OUT/cprof_fetch $OUT/cprof_write
cat $OUT/${TAG}_kernel_stats_coarse.txt | head -70
tail -c 1500 $OUT/${TAG}_bench_coarse.json
