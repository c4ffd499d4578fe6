#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r02o; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -30
python - <<'PY'
# row-major Gram: device matrix 1e8 x 16 f64 (12.8 GB) + y
import time, torch, numpy as np, ctypes as C, sys
sys.path.insert(0, ".")
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import _lib, lstsq
from polars_ds_extension_amd.linear_models import LR
n, p = 100_000_000, 16
X = torch.randn(n, p, dtype=torch.float64, device="cuda")
y = X @ torch.randn(p, dtype=torch.float64, device="cuda") + 0.1 * torch.randn(n, dtype=torch.float64, device="cuda")
ctx = lstsq.default_context(); ctx.follow_torch_stream(X.device)
lr = LR(); lr.fit(X, y)
ctx.set_timing(True); ctx.get_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): lr.fit(X, y)
torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 5
t = ctx.get_timing(True); ctx.set_timing(False)
ms = t["moments"][0] / t["moments"][1]
print(f"LR.fit on a row-major CUDA tensor 1e8 x 16 f64: wall {w*1e3:.3f} ms, Gram kernel {ms:.3f} ms = {n*(p+1)*8/ms/1e6:.0f} GB/s = {n*(p+1)*8/ms/1e6/8000:.3f} of HBM peak")
PY
