#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r04b; mkdir -p $O
LIB=polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
cp $LIB /tmp/default.so
cp tools/variants/prof.bin $LIB
timeout -k 5 200 python tools/rolling_pair_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_rolling.log
timeout -k 5 200 python tools/rolling_pair_profile.py expanding 2>&1 | grep -v amdgpu.ids | tee $O/prof_expanding.log
cp /tmp/default.so $LIB
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rolling or recursive" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
