#!/bin/bash
# round 2, call m: GPU suite, rolling / expanding after LDS-direct prefetch + streaming totals + parallel tile prefix, phases
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02m; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -12
echo "== rolling"
timeout -k 5 300 python tools/rolling_bench.py > $O/rolling.log 2>&1; grep -E "rolling|expanding" $O/rolling.log
echo "== rolling phases"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
cp tools/variants_prof/prof.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 200 python tools/rolling_seg_profile.py > $O/rolling_phase.log 2>&1; grep -v amdgpu $O/rolling_phase.log
cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o x -- python -u $GRAFT_REPO_ROOT/tools/rolling_bench.py c4 > $O/roll_stats_run.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/p3/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
