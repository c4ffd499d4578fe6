#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/keyed4; mkdir -p $O
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py tests/test_plugin_abi.py -m gpu -q -x -k "by_key or partition or c3_spec or pl_lr_by or order_check or pred" -p no:cacheprovider > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout -k 5 200 python tools/ab_quick.py keyed pred 2>&1 | grep -v amdgpu.ids | tail -4
