#!/bin/bash
# round 3, GPU call b: the whole -m gpu suite (multi-context by-key, partition route, fixed tests), partition route against the sorting
# route on one box (+ its stage trace), bench line (sliced pl_lr_by, pinned results), phase clocks of the rolling kernel
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r03b; mkdir -p $O
LIB=polars_ds_extension_amd/csrc/libpds_lstsq_hip.so
timeout -k 5 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout -k 5 300 python tools/ab_quick.py keyed pred > $O/keyed_partition.log 2>&1
PDS_KEYED_SORT=1 timeout -k 5 300 python tools/ab_quick.py keyed > $O/keyed_sort.log 2>&1
PDS_TRACE=1 timeout -k 5 300 python tools/ab_quick.py keyed > $O/keyed_trace.log 2>&1
timeout -k 5 700 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
cp $LIB /tmp/default_lib.so
cp tools/variants_prof/rollprof.bin $LIB
timeout -k 5 300 python tools/rolling_seg_profile.py > $O/rolling_phases.log 2>&1
cp /tmp/default_lib.so $LIB
echo "---- pytest"; tail -30 $O/pytest.log
echo "---- keyed partition"; grep -v amdgpu.ids $O/keyed_partition.log | tail -6
echo "---- keyed sort"; grep -v amdgpu.ids $O/keyed_sort.log | tail -3
echo "---- keyed trace"; grep "pds trace" $O/keyed_trace.log | tail -24
echo "---- bench"; python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03b/bench.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms_min_median_max"]); print(json.dumps(d["end_to_end"])); print(json.dumps(d["cpu_baseline"])[:1200]); print(json.dumps(d["grouped_c3spec"]))
else:
    print(open("gpurun_out/r03b/bench.log").read()[-2000:])
PY
echo "---- rolling phases"; grep -v amdgpu.ids $O/rolling_phases.log | tail -12
