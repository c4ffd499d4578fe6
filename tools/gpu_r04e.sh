#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r04e; mkdir -p $O
LIB=polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
cp $LIB /tmp/default.so; cp tools/variants/dev.bin $LIB
timeout -k 5 400 python tools/grouped_mid_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/phases.log
cp /tmp/default.so $LIB
timeout -k 5 300 python tools/grouped_mid_width.py 2>&1 | grep -v amdgpu.ids | tee $O/mid_width.log
