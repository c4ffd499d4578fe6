"""Per-phase shader-clock sums of rolling_pair_kernel (library built with EXTRA=-DPDS_PROFILE_ROLLING): python tools/rolling_pair_profile.py [expanding]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import polars_ds_extension_amd as pds  # noqa: E402
from polars_ds_extension_amd import _lib  # noqa: E402

n, p, w = 100_000_000, 8, 256
expanding = len(sys.argv) > 1 and sys.argv[1] == "expanding"
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
    if expanding:
        return pds.recursive_lin_reg(*xs, target=y, start_with=w, ctx=ctx)
    return pds.rolling_lin_reg(*xs, target=y, window_size=w, ctx=ctx)


f()
torch.cuda.synchronize()
so.pds_debug_rolling_cycles(buf, 1)
R = 3
for _ in range(R):
    f()
torch.cuda.synchronize()
so.pds_debug_rolling_cycles(buf, 1)
names = ["tile anchor", "waiting for the stage image", "pass 1 (chain increments, finiteness)", "rotate and scan (DPP)", "pass 2, rows 0-1",
         "pass 2, rows 2-3 (+ next stage's loads issued)", "-", "wave total"]
stages = n / 128
print(f"{'expanding' if expanding else 'rolling'}: wave-stages (128 rows) {stages:.3g}; clk per stage per wave {buf[7] / R / stages:.0f}")
for k, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:60s} {100.0 * buf[k] / buf[7]:5.1f} %   per stage {buf[k] / R / stages:8.0f}")
