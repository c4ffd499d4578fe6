#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/midgroup; mkdir -p $O
timeout -k 5 700 python -m pytest tests/test_gpu_parity.py tests/test_f32_contract.py tests/test_reference_suite.py -m gpu -q -x -k "grouped or group" -p no:cacheprovider > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout -k 5 200 python tools/grouped_mid_width.py 2>&1 | grep -v amdgpu
PDS_SOLVE_WAVE=0 timeout -k 5 200 python tools/grouped_mid_width.py 2>&1 | grep -v amdgpu | sed 's/^/[lds solver] /'
