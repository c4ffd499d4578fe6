#!/bin/bash
# round 2, call ap: 256 x 256 tile of the split arithmetic against the 128 x 128 tile (PDS_WIDE_TILE128=1) and the f32 instructions
mkdir -p gpurun_out
{
echo "== 256 tile"; timeout 600 python tools/wide_split_ab.py acc time
echo "== 128 tile"; PDS_WIDE_TILE128=1 timeout 600 python tools/wide_split_ab.py time
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_f32_contract.py tests/test_baseline_sizes.py -m gpu -x -q -k "wide or config5 or c5 or f32 or elastic or moments" 2>&1 | tail -5
} > gpurun_out/r02ap.log 2>&1
cat gpurun_out/r02ap.log
