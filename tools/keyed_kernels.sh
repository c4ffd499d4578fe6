#!/bin/bash
# per-kernel times of the shuffled-keys route under rocprofv3 (one variant of the library installed by tools/ab_variants.sh run)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $GRAFT_REPO_ROOT/tools/ab_quick.py keyed > /tmp/pk.log 2>&1
grep "keyed C3" /tmp/pk.log
python - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "part_scatter" in r["Name"] or "part_accum" in r["Name"]:
        print(f"   {r['Name'][30:80]:50s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
