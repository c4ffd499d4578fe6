#!/bin/bash
# scatter kernel of the keyed partition route against the bucket width (PDS_PART_SHIFT: ids per bucket = 2^shift)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/scatter_shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 6 7 8 9 10 11; do
  rm -rf /tmp/pk && PDS_PART_SHIFT=$d timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/run_$d.log 2>&1
  python - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" $d <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "part_scatter" in r["Name"] or "part_hist" in r["Name"] or "part_accum" in r["Name"]:
        print(f"shift {sys.argv[2]}: {r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}")
PY
done
