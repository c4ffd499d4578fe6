#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/keyed5; mkdir -p $O
timeout -k 5 700 python -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py tests/test_plugin_abi.py tests/test_polars_exprs.py -m gpu -q -x -k "by_key or partition or c3_spec or pl_lr_by or order_check or pred or run_lengths or multi" -p no:cacheprovider > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout -k 5 100 python tools/sorted_keys_prof.py 2>&1 | grep "ordered keys" | tail -2
timeout -k 5 200 python tools/ab_quick.py keyed 2>&1 | grep -v amdgpu.ids | tail -2
