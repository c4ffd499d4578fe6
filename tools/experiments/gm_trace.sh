#!/bin/bash
# On the GPU box: kernel trace of lin_reg_by at 24 features on the headline frame -- what runs beside the stream kernel and where the gaps are.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o t -- python $GRAFT_REPO_ROOT/tools/gm_ab.py p=${1:-24} > /tmp/gt.log 2>&1
f=$(find /tmp/gt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "pds::" in r["Kernel_Name"] or "hip" in r["Kernel_Name"].lower() or "memset" in r["Kernel_Name"].lower() or "fill" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last mid_empty_groups_kernel on
idx = max(i for i, r in enumerate(rows) if "mid_empty_groups" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:idx + 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"+{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:110]}")
    prev_end = e
PY
