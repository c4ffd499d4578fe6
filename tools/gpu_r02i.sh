#!/bin/bash
# round 2, call i: GPU suite (keyed record gather, rows_to_cols, chunked host frames), keyed timing + breakdown, bench line
# (end_to_end through the chunked staging), phase profile of the fused grouped kernel
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02i; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -25
echo "== keyed (records) vs by-column"
timeout -k 5 300 python tools/bench_extra.py keyed 2>/dev/null | grep -A3 wall_ms
PDS_KEYED_GATHER_BY_COLUMN=1 timeout -k 5 300 python tools/bench_extra.py keyed 2>/dev/null | grep -A3 wall_ms
echo "== bench line"
timeout -k 5 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i/bench_line.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","gram_build","grouped_p8","end_to_end","grouped_c3spec"): print(k, json.dumps(d.get(k)))
PY
echo "== grouped phases"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
cp tools/variants_prof/phases.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 200 python tools/phase_profile.py > $O/grouped_phases.log 2>&1; grep -v amdgpu $O/grouped_phases.log
P=8 timeout -k 5 200 python tools/phase_profile.py > $O/grouped_phases_p8.log 2>&1; grep -v amdgpu $O/grouped_phases_p8.log
cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
echo "== keyed breakdown (records)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $GRAFT_REPO_ROOT/tools/bench_extra.py keyed > $O/keyed_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02i/keyed_kernel_stats.csv")))
for r in rows[:14]: print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
