#!/bin/bash
# round-2 GPU call A: full GPU suite (new BASELINE-size tests included), f32 distance table, membw sweep, bench line
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r02a; mkdir -p $O
timeout -k 5 1000 python -m pytest tests -m gpu -q --maxfail=40 -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout -k 5 700 python tools/f32_distances.py big > $O/f32.log 2>&1; echo "rc=$?" >> $O/f32.log
timeout -k 5 120 tools/membw.bin > $O/membw.log 2>&1
timeout -k 5 300 python bench.py > $O/bench.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -40
grep -v amdgpu.ids $O/f32.log | tail -50
tail -30 $O/membw.log
tail -3 $O/bench.log
