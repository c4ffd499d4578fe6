#!/bin/bash
# round 2, call k: GPU suite (rolling beyond 64 coefficients), keyed timing after the record-gather change
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02k; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -40
echo "== keyed"
timeout -k 5 300 python tools/bench_extra.py keyed 2>/dev/null | grep -A3 wall_ms
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $GRAFT_REPO_ROOT/tools/bench_extra.py keyed > $O/keyed_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02k/keyed_kernel_stats.csv")))
for r in rows[:8]: print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
