#!/bin/bash
# by-key parity tests + C3 shuffled-keys wall and per-kernel times (no PMC passes) + the mid-width report tests
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/keyed3; mkdir -p $O
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py -m gpu -q -k "by_key or partition or c3_spec or pl_lr_by or wide_weighted_and_hc" -p no:cacheprovider > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/keyed_prof_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
cd $ROOT
echo "---- pytest"; tail -8 $O/pytest.log
echo "---- keyed"; grep -v amdgpu.ids $O/keyed_prof_run.log | tail -3
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/keyed3/keyed_kernel_stats.csv")):
    if float(r["TotalDurationNs"]) > 2e6:
        print(f"{r['Name'][:110]:110s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
