"""Development aid: fused grouped kernel with an intercept (centred form) at 16 and 8 features."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
G, R = 1_000_000, 100
N = G * R
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(16)]
y = sum(x * 0.1 for x in xs) + 0.5 + 0.1 * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
for P in (16, 8):
    for bias in (False, True):
        f = lambda: pds.lin_reg_by(*xs[:P], target=y, group_offsets=off, add_bias=bias, ctx=ctx)
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): co, nu = f()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
        print(f"p={P} bias={bias}: {wall*1e3:.3f} ms/step, nulls {int(nu.sum())}, bias mean {float(co[:, -1].mean()):.4f}")
