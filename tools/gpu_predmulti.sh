#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/predmulti; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_plugin_abi.py -m gpu -q -x -k "multi_context or sliced_route or by_pred" -p no:cacheprovider > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout -k 5 600 python bench.py --no-cpu --steps 3 --warmup 1 > $O/bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/predmulti/bench.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(json.dumps(d["end_to_end"])[:1600])
else:
    print(open("gpurun_out/predmulti/bench.log").read()[-1500:])
PY
