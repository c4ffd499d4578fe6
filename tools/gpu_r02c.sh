#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
python tools/grouped_second_pass_cost.py 2>&1 | grep -v amdgpu.ids
echo "=== r01 library"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/cur.so
cp tools/variants/r01.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
python tools/grouped_second_pass_cost.py 2>&1 | grep -v amdgpu.ids
cp /tmp/cur.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
