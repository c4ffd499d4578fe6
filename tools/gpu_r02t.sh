#!/bin/bash
# round 2, final evidence call: full GPU suite, smoke, the round's profile set
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02t; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -12
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/profile_round.sh > $O/profile_round.log 2>&1
head -c 5000 gpurun_out/prof/bench_line.json
