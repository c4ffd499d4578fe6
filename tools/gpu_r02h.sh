#!/bin/bash
# round 2, call h: GPU suite, report / residual-pass timings, rolling variants A/B + phase profile, keyed-path breakdown
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02h; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -25
echo "== report / single (default build)"
timeout -k 5 300 python tools/bench_extra.py report single > $O/report.json 2> $O/report.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r02h/report.json"))
for k,v in d.items(): print(k, json.dumps(v))
PY
echo "== same with PDS_REPORT_NO_FUSE=1"
PDS_REPORT_NO_FUSE=1 timeout -k 5 300 python tools/bench_extra.py report 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items(): print(k, json.dumps(v))"
echo "== rolling variants"
bash tools/ab_variants.sh run "python tools/rolling_bench.py c4" 1 > $O/rolling_ab.log 2>&1; grep -E "variant|rolling|expanding" $O/rolling_ab.log
echo "== rolling phase profile"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
cp tools/variants_prof/prof.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 200 python tools/rolling_seg_profile.py > $O/rolling_phase.log 2>&1; grep -v amdgpu $O/rolling_phase.log
cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
echo "== keyed breakdown"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $GRAFT_REPO_ROOT/tools/bench_extra.py keyed > $O/keyed_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02h/keyed_kernel_stats.csv")))
for r in rows[:24]: print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
