#!/bin/bash
# fused grouped kernel: two tiles of register prefetch per wave (PDS_GROUPED_PF2=1) against one, same box
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/pf2; mkdir -p $O
PDS_GROUPED_PF2=1 timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py -m gpu -q -x -k "grouped or headline or c3_spec" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for r in 1 2; do
  for v in 0 1; do
    echo "== round $r pf2 $v"; PDS_GROUPED_PF2=$v timeout -k 5 200 python tools/ab_quick.py grouped 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
