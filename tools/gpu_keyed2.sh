#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/keyed2; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py -m gpu -q -k "by_key or partition or c3_spec" -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for a in 0 1; do
  rm -rf /tmp/pk && PDS_PART_DEBUG=$a timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/run_$a.log 2>&1
  grep "keyed C3" $O/run_$a.log
  python - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" $a <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("part_" in r["Name"] or "solve_reg" in r["Name"] or "key_order" in r["Name"]) and float(r["TotalDurationNs"]) > 2e5:
        print(f"debug={sys.argv[2]}: {r['Name'][30:75]} avg {float(r['AverageNs']) / 1e3:.1f} us x {r['Calls']}")
PY
done
