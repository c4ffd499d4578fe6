"""Per-phase shader-clock sums of the lane = K rows rolling kernel (library built with EXTRA=-DPDS_PROFILE_ROLLING)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import polars_ds_extension_amd as pds  # noqa: E402
from polars_ds_extension_amd import _lib  # noqa: E402

n, p, w = 100_000_000, 8, 256
dev = torch.device("cuda", 0)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev)
gen.manual_seed(3)
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
y = sum(xs[j] * (0.1 * (j + 1)) for j in range(p)) + 1e-3 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
so = _lib.load()
buf = (C.c_ulonglong * 8)()


def f():
    return pds.rolling_lin_reg(*xs, target=y, window_size=w, ctx=ctx)


f()
torch.cuda.synchronize()
so.pds_debug_rolling_cycles(buf, 1)
R = 3
for _ in range(R):
    f()
torch.cuda.synchronize()
so.pds_debug_rolling_cycles(buf, 1)
names = ["anchor + first stage of a tile", "rows LDS -> registers, finiteness", "pass 1 (lane's increments)", "scan through LDS", "pass 2 (4 rows: update, solve, store)",
         "  of which: waiting for / committing the prefetched pieces", "-", "wave total"]
stages = n / 256
print(f"stages (256 rows) {stages:.3g}; clk per stage per wave {buf[7] / R / stages:.0f}")
for k, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:60s} {100.0 * buf[k] / buf[7]:5.1f} %   per stage {buf[k] / R / stages:8.0f}")
