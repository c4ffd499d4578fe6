#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02n; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "rolling or recursive or windowed or c4 or f32 or reference_suite or online or polars" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -6
timeout -k 5 300 python tools/rolling_bench.py > $O/rolling.log 2>&1; grep -E "rolling|expanding" $O/rolling.log
