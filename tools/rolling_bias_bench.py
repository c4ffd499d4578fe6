"""Development aid: rolling kernel, 7 features + bias (p' = 8) against 8 features without."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
n, w = 100_000_000, 256
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=g) for _ in range(8)]
y = sum(xs[j] * (0.1 * (j + 1)) for j in range(8)) + 0.3 + 1e-3 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)
for p, bias in ((8, False), (7, True), (5, True), (3, True)):
    pds.rolling_lin_reg(*xs[:p], target=y, window_size=w, add_bias=bias, ctx=ctx)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    for _ in range(3): pds.rolling_lin_reg(*xs[:p], target=y, window_size=w, add_bias=bias, ctx=ctx)
    ctx.set_timing(False); t = ctx.get_timing(reset=True)["rolling"]
    print(f"p={p} bias={bias}: {t[0] / t[1]:.3f} ms")
