#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r04c; mkdir -p $O
LIB=polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
cp $LIB /tmp/default.so
for v in noold base; do cp tools/variants/$v.bin $LIB; echo "== $v"; timeout -k 5 200 python tools/rolling_bench.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/bench.log; done
cp tools/variants/noold_prof.bin $LIB
timeout -k 5 200 python tools/rolling_pair_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_noold.log
cp /tmp/default.so $LIB
