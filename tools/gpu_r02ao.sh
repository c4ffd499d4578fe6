#!/bin/bash
# round 2, call ao: the default bench run (wall clock of the whole run) with the C5 leg
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py > gpurun_out/r02ao_bench.json 2> gpurun_out/r02ao_bench.err
echo "bench.py default run: $(( $(date +%s) - t0 )) s wall, rc=$?"
tail -2 gpurun_out/r02ao_bench.err
python tools/bench_brief.py < gpurun_out/r02ao_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02ao_bench.json"))
print(json.dumps(d["other_configs"], indent=1))
PY
