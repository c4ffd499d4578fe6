#!/bin/bash
# round 2, call l: rolling with the direct-to-LDS prefetch (A/B + parity of that build)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02l; mkdir -p $O
bash tools/ab_variants.sh run "python tools/rolling_bench.py" 2 > $O/rolling_ab.log 2>&1; grep -E "variant|rolling|expanding" $O/rolling_ab.log
echo "== parity of the LDS-direct build"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
cp tools/variants/ldsd.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "rolling or recursive or windowed or c4 or f32 or reference_suite or online" > $O/pytest_ldsd.log 2>&1; echo "rc=$?" >> $O/pytest_ldsd.log
grep -v amdgpu.ids $O/pytest_ldsd.log | tail -8
cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
