#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects everything profiles/rNN_* is built from into gpurun_out/prof/.
#   bash tools/profile_round.sh        then, back in the repo:  python tools/summarize_profiles.py r06
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
# Every profiler run is under `timeout` and writes to a file (tools/README.md, GPU-box hygiene).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# counter / trace CSVs carry every torch kernel of the frame generators: keep the header and the library's kernels only (gpurun merges
# at most 64 MiB back; round 5 lost a whole call's output to that limit)
slim() { for f in "$@"; do [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep "pds::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }; done; }
timeout -k 5 600 python -u $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
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
# round 4: HBM counters of the mid-width kernels (moments_mid, leverage_mid, fused report), the grouped 17 .. 32-feature stream and the
# grouped pred pass; the 16 -> 17 feature cliff; matrix-pipe / wait-state counters of the headline kernel with the Gram kernel as control
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p8 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p8 -o w -- python -u $ROOT/tools/wide_report_prof.py > $OUT/pmc_wide_$c.log 2>&1
  cp $(find /tmp/p8 -name "*counter_collection.csv" | head -1) $OUT/pmc_wide_$c.csv
  rm -rf /tmp/p9 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p9 -o g -- python -u $ROOT/tools/grouped_mid_width.py > $OUT/pmc_gmid_$c.log 2>&1
  cp $(find /tmp/p9 -name "*counter_collection.csv" | head -1) $OUT/pmc_gmid_$c.csv
  rm -rf /tmp/p10 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p10 -o q -- python -u $ROOT/tools/ab_quick.py pred > $OUT/pmc_pred_$c.log 2>&1
  cp $(find /tmp/p10 -name "*counter_collection.csv" | head -1) $OUT/pmc_pred_$c.csv
done
timeout -k 5 300 python -u $ROOT/tools/grouped_width_cliff.py > $OUT/width_cliff.log 2>&1
PDS_GROUPED_MID_FUSED=0 timeout -k 5 300 python -u $ROOT/tools/grouped_width_cliff.py >> $OUT/width_cliff.log 2>&1
k=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  k=$((k+1))
  rm -rf /tmp/p11 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p11 -o s -- python -u $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pmc_sq_$k.log 2>&1
  f=$(find /tmp/p11 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pmc_sq_$k.csv
done
# round 5: the partition route's id-indexed prediction pass (grouped_pred MODE 3), the C5 wide Gram's counters (FETCH / WRITE, L2 hit rate,
# matrix-pipe busy cycles), the ordered-keys call's kernels, the HC2 / HC3 report widths
bash $ROOT/tools/pmc_wide_r05.sh > $OUT/pmc_wide_r05.log 2>&1; cp $ROOT/gpurun_out/pmc_wide/r05_pmc_wide.json $OUT/pmc_wide.json 2>/dev/null; rm -rf $ROOT/gpurun_out/pmc_wide
cd /tmp
rm -rf /tmp/p12 && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p12 -o k -- python -u $ROOT/tools/sorted_keys_ab.py > $OUT/sorted_keys_run.log 2>&1
cp $(find /tmp/p12 -name "*kernel_stats.csv" | head -1) $OUT/sorted_keys_kernel_stats.csv
timeout -k 5 300 python -u $ROOT/tools/report_hc_quick.py > $OUT/report_hc.log 2>&1
# round 6: the HC2 / HC3 report route at C2 (1e8 x 16 / 12 / 9 / 8 f64 + intercept): kernel stats and HBM counters of its kernels
# (moments_small_kernel<double, 4, ...>: residuals + leverages + meat fused), the two-process direct gather smoke
rm -rf /tmp/p13 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p13 -o h -- python -u $ROOT/tools/report_hc_quick.py > $OUT/report_hc_run.log 2>&1
cp $(find /tmp/p13 -name "*kernel_stats.csv" | head -1) $OUT/hc_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p14 && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p14 -o h -- python -u $ROOT/tools/report_hc_quick.py > $OUT/pmc_hc_$c.log 2>&1
  cp $(find /tmp/p14 -name "*counter_collection.csv" | head -1) $OUT/pmc_hc_$c.csv
done
slim $OUT/pmc_*.csv $OUT/bench_kernel_trace.csv
du -sh $OUT
ls -la $OUT
