#!/bin/bash
# round 2, call j: GPU suite (GLM), A/B of the fused grouped kernel variants (spread load issue, half tile = 3 waves / SIMD),
# parity of the half-tile build
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02j; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -25
echo "== fused grouped kernel variants"
bash tools/ab_variants.sh run "python bench.py --no-cpu --no-extras --steps 20 --warmup 5 | python tools/bench_brief.py" 2 > $O/grouped_ab.log 2>&1
grep -E "variant|step " $O/grouped_ab.log
echo "== parity of the half-tile build"
cp polars_ds_extension_amd/csrc/libpds_lstsq_hip.so /tmp/keep.so
cp tools/variants/half.bin polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "grouped or lr_by or c3 or fused or by_key or smoke or baseline or f32" > $O/pytest_half.log 2>&1; echo "rc=$?" >> $O/pytest_half.log
grep -v amdgpu.ids $O/pytest_half.log | tail -8
cp /tmp/keep.so polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
