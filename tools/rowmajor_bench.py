"""LR.fit on a row-major CUDA tensor (pds_lr_rowmajor_*, moments_rowmajor_f64_kernel): Gram kernel time at 1e8 x 16 f64."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from polars_ds_extension_amd import lstsq  # noqa: E402
from polars_ds_extension_amd.linear_models import LR  # noqa: E402

n = 100_000_000
for p in (16, 8):
    X = torch.randn(n, p, dtype=torch.float64, device="cuda")
    y = X @ torch.randn(p, dtype=torch.float64, device="cuda") + 0.1 * torch.randn(n, dtype=torch.float64, device="cuda")
    ctx = lstsq.default_context()
    ctx.follow_torch_stream(X.device)
    lr = LR()
    lr.fit(X, y)
    ctx.set_timing(True)
    ctx.get_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lr.fit(X, y)
    torch.cuda.synchronize()
    w = (time.perf_counter() - t0) / 5
    t = ctx.get_timing(True)
    ctx.set_timing(False)
    ms = t["moments"][0] / t["moments"][1]
    gb = n * (p + 1) * 8 / 1e9
    print(f"row-major 1e8 x {p} f64: wall {w * 1e3:.3f} ms, Gram kernel {ms:.3f} ms = {gb / ms * 1e3:.0f} GB/s = {gb / ms * 1e3 / 8000:.3f} of HBM peak", flush=True)
    del X, y
    torch.cuda.empty_cache()
