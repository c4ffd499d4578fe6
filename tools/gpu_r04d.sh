#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r04d; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "grouped_mid or grouped_17 or mid_width" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout -k 5 300 python tools/grouped_width_cliff.py 2>&1 | grep -v amdgpu.ids | tee $O/cliff_fused.log
PDS_GROUPED_MID_FUSED=0 timeout -k 5 300 python tools/grouped_width_cliff.py 2>&1 | grep -v amdgpu.ids | tee $O/cliff_records.log
timeout -k 5 300 python tools/grouped_mid_width.py 2>&1 | grep -v amdgpu.ids | tee $O/mid_width.log
