#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD; O=$PWD/gpurun_out/wide; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -q -k "moments or wide or multi_target or report or glm or GLM or beyond or hc" -p no:cacheprovider > $O/pytest.log 2>&1; tail -6 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o w -- python -u $ROOT/tools/wide_report_prof.py > $O/run.log 2>&1
grep "n=" $O/run.log
python - "$(find /tmp/pw -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pds::" in r["Name"] and float(r["TotalDurationNs"]) > 3e6:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
