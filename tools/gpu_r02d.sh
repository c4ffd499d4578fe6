#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02d; mkdir -p $O
export PDS_PROBE_ONLY=8
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp && timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/grouped_second_pass_cost.py > $O/run.log 2>&1
f=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $O/trace_summary.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-260:]:
    name = r["Kernel_Name"][:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{name:70s} dur {(e - s) / 1e3:9.1f} us   gap_before {gap:9.1f} us")
    prev_end = e
PY
grep -v amdgpu.ids $O/run.log | tail -12
tail -70 $O/trace_summary.txt
