#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects everything profiles/rNN_* is built from into gpurun_out/prof/.
#   bash tools/profile_round.sh        then, back in the repo:  python tools/summarize_profiles.py r03
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
# Every profiler run is under `timeout` and writes to a file (tools/README.md, GPU-box hygiene).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python -u $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
# same flags as the bench line's timed region (warm-ups excluded from the averages by summarize_profiles.py: it drops the
# first `warmup` launches of each kernel using the kernel trace, not the --stats table)
rm -rf /tmp/p1 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python -u $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-extras > $OUT/stats_run.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
cp $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) $OUT/bench_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p2 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p2 -o b -- python -u $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/pmc_$c.log 2>&1
  cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $OUT/pmc_$c.csv
  rm -rf /tmp/p4 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p4 -o r -- python -u $ROOT/tools/rolling_bench.py c4 > $OUT/pmc_roll_$c.log 2>&1
  cp $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $OUT/pmc_roll_$c.csv
done
timeout -k 5 500 python -u $ROOT/tools/bench_extra.py > $OUT/bench_extra.json 2> $OUT/bench_extra.err
rm -rf /tmp/p3 && timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o x -- python -u $ROOT/tools/bench_extra.py rolling en single report > $OUT/extra_stats_run.log 2>&1
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $OUT/extra_kernel_stats.csv
# round 3: the routes added this round -- shuffled keys (partition route), grouped pred, mid-width reports -- kernel stats and HBM counters
rm -rf /tmp/p5 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o k -- python -u $ROOT/tools/ab_quick.py keyed pred > $OUT/keyed_run.log 2>&1
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $OUT/keyed_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p6 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p6 -o k -- python -u $ROOT/tools/ab_quick.py keyed > $OUT/pmc_keyed_$c.log 2>&1
  cp $(find /tmp/p6 -name "*counter_collection.csv" | head -1) $OUT/pmc_keyed_$c.csv
done
rm -rf /tmp/p7 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -o w -- python -u $ROOT/tools/wide_report_prof.py > $OUT/wide_run.log 2>&1
cp $(find /tmp/p7 -name "*kernel_stats.csv" | head -1) $OUT/wide_kernel_stats.csv
# grouped fits with 17 .. 64 features: the wave-per-system register solver against the LDS solver (PDS_SOLVE_WAVE=0)
timeout -k 5 200 python -u $ROOT/tools/grouped_mid_width.py > $OUT/grouped_mid.log 2>&1
PDS_SOLVE_WAVE=0 timeout -k 5 300 python -u $ROOT/tools/grouped_mid_width.py 2>&1 | sed 's/^/[PDS_SOLVE_WAVE=0: LDS pivoted QR for every system] /' >> $OUT/grouped_mid.log
timeout -k 5 100 python -u $ROOT/tools/sorted_keys_prof.py >> $OUT/keyed_run.log 2>&1
ls -la $OUT
