#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/pred; mkdir -p $O
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py tests/test_plugin_abi.py tests/test_polars_exprs.py tests/test_baseline_sizes.py -m gpu -q -x -k "pred or rolling or recursive" -p no:cacheprovider > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout -k 5 200 python tools/ab_quick.py pred rolling 2>&1 | grep -v amdgpu.ids | tail -6
