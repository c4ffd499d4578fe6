#!/bin/bash
# round 2, closing call: full GPU suite, smoke, the launcher path of bench.py at world 1 (RCCL init, gather, scatter leg), profile set
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02z; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -8
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench.py under torch.distributed.run, world 1 (strong scaling path: gather inside the timed region, scatter leg)"
PDS_BENCH_FORCE_DIST=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_dist1.json 2> $O/bench_dist1.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r02z/bench_dist1.json") if l.startswith("{")][-1])
    print({k: d[k] for k in ("value","ms_per_step","scaling","n_gpus")}, d["config"]["parallelism"], "scatter:", d.get("scatter"))
except Exception as e:
    print("dist bench failed:", e); print(open("gpurun_out/r02z/bench_dist1.err").read()[-1500:])
PY
bash tools/profile_round.sh > $O/profile_round.log 2>&1
head -c 1500 gpurun_out/prof/bench_line.json; echo
