#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects everything profiles/rNN_* is built from into gpurun_out/prof/.
#   bash tools/profile_round.sh        then, back in the repo:  python tools/summarize_profiles.py r01
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -u $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python -u $ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/stats_run.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p2 && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p2 -o b -- python -u $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_$c.log 2>&1
  cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $OUT/pmc_$c.csv
done
python -u $ROOT/tools/bench_extra.py > $OUT/bench_extra.json 2> $OUT/bench_extra.err
rm -rf /tmp/p3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o x -- python -u $ROOT/tools/bench_extra.py rolling en single report > $OUT/extra_stats_run.log 2>&1
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $OUT/extra_kernel_stats.csv
ls -la $OUT
