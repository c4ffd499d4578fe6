#!/bin/bash
# round 3, GPU call c: v_fma_f64 issue rate / latency per wave, rolling wait placement A/B, per-kernel times of the partition route
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$PWD
O=$PWD/gpurun_out/r03c; mkdir -p $O
(timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu tools/valu_f64_rate.hip 2>/dev/null && timeout 120 /tmp/valu) > $O/valu.log 2>&1
timeout -k 5 600 bash tools/ab_variants.sh run "python tools/ab_quick.py rolling" 2 > $O/ab.log 2>&1
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_context or partition" -p no:cacheprovider > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python -u $ROOT/tools/ab_quick.py keyed > $O/keyed_prof_run.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $O/keyed_kernel_stats.csv
cd $ROOT
echo "---- valu"; cat $O/valu.log
echo "---- ab rolling"; grep -v amdgpu.ids $O/ab.log | tail -16
echo "---- pytest"; tail -5 $O/pytest.log
echo "---- keyed kernels"; head -25 $O/keyed_kernel_stats.csv | cut -c1-220
