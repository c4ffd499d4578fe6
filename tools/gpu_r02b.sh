#!/bin/bash
# round-2 GPU call B: A/B of the streaming-load variants (one box), then the BASELINE-size tests
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r02b; mkdir -p $O
bash tools/ab_variants.sh run "python bench.py --no-cpu --steps 10 --warmup 3 | python tools/bench_brief.py" 2 > $O/ab.log 2>&1
timeout -k 5 600 python -m pytest tests/test_baseline_sizes.py -m gpu -q -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/ab.log
grep -v amdgpu.ids $O/pytest.log | tail -15
