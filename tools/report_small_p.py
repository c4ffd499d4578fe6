"""Development aid: lin_reg_report wall time at 1e8 rows for a few feature counts."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
n = 100_000_000
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=g) for _ in range(8)]
y = sum(xs[j] * (0.1 * (j + 1)) for j in range(8)) + 0.3 + 1e-2 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)
yv = float(y.var())
for p in (1, 2, 4, 8):
    f = lambda: pds.lin_reg_report(*xs[:p], target=y, add_bias=True, y_var=yv, ctx=ctx)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
    print(f"p={p}: report {wall*1e3:.3f} ms  ({2 * n * (p + 1) * 8 / wall / 1e9:.0f} GB/s over two passes)")
