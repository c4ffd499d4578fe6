#!/bin/bash
# the whole -m gpu suite + the bench line (what the driver runs at round end)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/full; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout -k 5 700 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
echo "---- pytest"; tail -12 $O/pytest.log
echo "---- smoke"; tail -2 $O/smoke.log
echo "---- bench"; python - <<'PY'
import json
l=[x for x in open("gpurun_out/full/bench.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_min_median_max"]); print(json.dumps(d["end_to_end"])[:900]); print(json.dumps(d["grouped_c3spec"])); print(json.dumps(d["other_configs"])[:1500])
else:
    print(open("gpurun_out/full/bench.log").read()[-2000:])
PY
