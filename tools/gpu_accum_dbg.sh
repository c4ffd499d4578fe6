#!/bin/bash
# accumulate kernel of the keyed partition route: time with parts of it switched off (PDS_PART_DEBUG: 1 no owners' loop, 2 stream only)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/accum_dbg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2; do
  rm -rf /tmp/pk && PDS_PART_DEBUG=$d timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/run_$d.log 2>&1
  python - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" $d <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "part_" in r["Name"]:
        print(f"debug {sys.argv[2]}: {r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
