#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02p; mkdir -p $O
bash tools/ab_variants.sh run "python tools/rowmajor_bench.py" 1 > $O/rowmajor_ab.log 2>&1; grep -E "variant|row-major" $O/rowmajor_ab.log
echo "== bench line (other_configs)"
timeout -k 5 400 python bench.py --no-cpu > $O/bench_line.json 2> $O/bench_line.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02p/bench_line.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","other_configs"): print(k, json.dumps(d.get(k)))
PY
