"""Development aid: per-phase shader-clock sums of the rolling kernel (build with EXTRA=-DPDS_PROFILE_ROLLING)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import _lib
n, p, w = 100_000_000, 8, 256
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
y = sum(xs[j] * (0.1 * (j + 1)) for j in range(p)) + 1e-3 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
so = _lib.load()
buf = (C.c_ulonglong * 8)()
f = lambda: pds.rolling_lin_reg(*xs, target=y, window_size=w, ctx=ctx)
f(); torch.cuda.synchronize(); so.pds_debug_rolling_cycles(buf, 1)
K = 3
for _ in range(K): f()
torch.cuda.synchronize(); so.pds_debug_rolling_cycles(buf, 1)
names = ["row hand-over (wait rows, isfinite, issue next)", "A: increments -> LDS", "B: scan", "C: read back + solve + store", "-", "-", "-", "wave total"]
steps = n / 64 * (1 + 256 / 4096)
print(f"steps (64 rows, incl. warm-up) {steps:.3g}; clk per step per wave {buf[7] / K / steps:.0f}")
for k, nm in enumerate(names):
    if nm != "-": print(f"  {nm:50s} {100.0 * buf[k] / buf[7]:5.1f} %   per step {buf[k] / K / steps:8.0f}")
print(f"  {'unaccounted':50s} {100.0 * (buf[7] - sum(buf[:4])) / buf[7]:5.1f} %")
