#!/bin/bash
# round 4, first GPU call: the pair kernel's parity tests + C4 timing, A/B against the lane = 4 rows kernel
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r04a; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rolling or recursive" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout -k 5 300 python tools/rolling_bench.py c4 > $O/bench_pair.log 2>&1; cat $O/bench_pair.log
PDS_ROLL_PAIR=0 timeout -k 5 300 python tools/rolling_bench.py c4 > $O/bench_seg.log 2>&1; cat $O/bench_seg.log
