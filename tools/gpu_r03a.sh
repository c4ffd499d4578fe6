#!/bin/bash
# round 3, GPU call a: new -m gpu tests (headline / C5 at size, grouped pred, null keys, widened coalescer), the bench line with the
# performance-build cpu_baseline, v_rcp_f64 accuracy, A/B of the rolling (register-resident leaving rows, tile, Newton, pair) and
# grouped (Newton) kernel variants on one box
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r03a; mkdir -p $O
(timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/rcp tools/rcp_accuracy.hip && timeout 60 /tmp/rcp) > $O/rcp.log 2>&1
timeout -k 5 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout -k 5 700 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
timeout -k 5 900 bash tools/ab_variants.sh run "python tools/ab_quick.py rolling grouped" 1 > $O/ab.log 2>&1
PDS_ROLL_OLD_STREAM=1 timeout -k 5 200 python tools/ab_quick.py rolling > $O/ab_oldstream.log 2>&1
timeout -k 5 300 python tools/ab_quick.py pred keyed > $O/ab_pred.log 2>&1
echo "---- rcp"; cat $O/rcp.log
echo "---- pytest"; tail -25 $O/pytest.log
echo "---- bench"; tail -3 $O/bench.log | cut -c1-3000
echo "---- ab"; grep -v "amdgpu.ids" $O/ab.log | tail -40
echo "---- old stream"; grep -v "amdgpu.ids" $O/ab_oldstream.log | tail -5
echo "---- pred / keyed"; grep -v "amdgpu.ids" $O/ab_pred.log | tail -8
