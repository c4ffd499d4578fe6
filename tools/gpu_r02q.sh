#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02q; mkdir -p $O
bash tools/ab_variants.sh run "python tools/rolling_bench.py" 2 > $O/rolling_ab.log 2>&1; grep -E "variant|rolling|expanding" $O/rolling_ab.log
for v in single pair; do
  echo "== parity of $v"
  cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
  cp tools/variants/$v.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
  timeout -k 5 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "rolling or recursive or windowed or c4 or f32 or reference_suite or online or polars or report" > $O/pytest_$v.log 2>&1; echo "rc=$?" >> $O/pytest_$v.log
  grep -v amdgpu.ids $O/pytest_$v.log | tail -4
  cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
done
