#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r02f; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -x -k "rolling or recursive or windowed or c4" > $O/pytest_roll.log 2>&1; echo "rc=$?" >> $O/pytest_roll.log
grep -v amdgpu.ids $O/pytest_roll.log | tail -30
echo "== rolling v2"; timeout 200 python tools/rolling_bench.py 2>&1 | grep -v amdgpu
echo "== rolling v1"; PDS_ROLLING_V1=1 timeout 200 python tools/rolling_bench.py 2>&1 | grep -v amdgpu
export PDS_PROBE_ONLY=8
echo "== second-pass probe"; python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -30
