#!/bin/bash
# round 2, call g: full GPU suite (incl. the f32 contract table), the round's profile set, A/B of the double-buffered wide Gram
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r02g; mkdir -p $O
timeout -k 5 240 python -m pytest tests/test_f32_contract.py -m gpu -q -s -p no:cacheprovider > $O/f32_contract.log 2>&1; echo "rc=$?" >> $O/f32_contract.log
grep -v amdgpu.ids $O/f32_contract.log | grep -E "gpu-truth|passed|failed|rc=|Error|assert" | head -80
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --deselect tests/test_f32_contract.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -15
bash tools/profile_round.sh > $O/profile_round.log 2>&1
cat gpurun_out/prof/bench_line.json | head -c 6000
echo "== A/B wide Gram double buffer"
bash tools/ab_variants.sh run "python tools/bench_extra.py en" 1 2>&1 | tail -12
