#!/bin/bash
# fused grouped kernel: three-wave workgroups with a solver wave (PDS_GROUPED_SPLIT=1) against the one-wave form, same box
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/split; mkdir -p $O
PDS_GROUPED_SPLIT=1 timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py -m gpu -q -x -k "grouped or headline or c3_spec or by_key" -p no:cacheprovider > $O/pytest_split.log 2>&1
tail -6 $O/pytest_split.log
for r in 1 2; do
  for v in 0 1; do
    echo "== round $r split $v"; PDS_GROUPED_SPLIT=$v timeout -k 5 200 python tools/ab_quick.py grouped 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
