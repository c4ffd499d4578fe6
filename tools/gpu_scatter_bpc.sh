#!/bin/bash
# scatter kernel of the keyed partition route against its blocks per CU (PDS_PART_SCATTER_BPC)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/scatter_bpc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 4 2 6 8 4; do
  rm -rf /tmp/pk && PDS_PART_SCATTER_BPC=$d timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/run_$d.log 2>&1
  python - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" $d <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "part_scatter" in r["Name"] or "solve_reg" in r["Name"] or "minmax" in r["Name"]:
        print(f"bpc {sys.argv[2]}: {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f}")
PY
  grep "keyed C3" $O/run_$d.log
done
