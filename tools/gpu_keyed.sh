#!/bin/bash
# per-kernel times of the keyed (shuffled rows) routes + the by-key parity tests
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/keyed; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py -m gpu -q -k "by_key or partition or c3_spec or pl_lr_by" -p no:cacheprovider > $O/pytest.log 2>&1
timeout -k 5 200 python tools/ab_quick.py keyed > $O/keyed.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/keyed_prof_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
cd $ROOT
echo "---- pytest"; tail -5 $O/pytest.log
echo "---- keyed"; grep -v amdgpu.ids $O/keyed.log | tail -3
echo "---- kernels"; grep "pds::" $O/keyed_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150 | head -14
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pp -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/pmc_$c.log 2>&1
  cp $(find /tmp/pp -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
done
cd $ROOT
python - <<'PY'
import csv, collections
for c, f in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/keyed/pmc_{c}.csv")):
        if "pds::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]) * f)
    for k, v in acc.items():
        if sum(v) / len(v) > 1e8:
            print(f"{c:11s} {k:70s} {sum(v) / len(v) / 1e9:8.2f} GB per launch ({len(v)} launches)")
PY
