"""Development aid: the keyed (shuffled rows) path at several frame sizes -- per-row cost of the gather against the size of the
frame it reads at random (is it the DRAM, or the address translation / caches in front of it?).  Run on the GPU box."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds

dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(5)
p, R = 8, 100
for G in (20_000, 100_000, 300_000, 1_000_000):
    N = G * R
    xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs) + 0.1 * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
    keys = torch.arange(G, device=dev, dtype=torch.int64).repeat_interleave(R)
    perm = torch.randperm(N, device=dev, generator=gen)
    xs_s = [x[perm] for x in xs]; y_s = y[perm]; k_s = keys[perm]
    f = lambda: pds.lin_reg_by_key(*xs_s, target=y_s, key=k_s, ctx=ctx, max_groups=G)
    f(); torch.cuda.synchronize()
    ctx.get_timing(True); ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    ctx.set_timing(False)
    t = {k: round(ms / max(c, 1), 3) for k, (ms, c) in ctx.get_timing(True).items() if c}
    print(f"G={G:8d} rows={N:.0e} frame={N * (p + 1) * 8 / 1e9:.2f} GB  wall {wall * 1e3:7.2f} ms = {wall / N * 1e9:.3f} ns/row   kernel kinds {t}")
    del xs, y, keys, perm, xs_s, y_s, k_s
    torch.cuda.empty_cache()
