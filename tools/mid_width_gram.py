import sys, time
sys.path.insert(0, '.')
import torch
import polars_ds_extension_amd as pds
dev = torch.device('cuda', 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
for n, p, dt in ((20_000_000, 64, torch.float32), (20_000_000, 32, torch.float32), (20_000_000, 32, torch.float64), (20_000_000, 100, torch.float64), (20_000_000, 20, torch.float64)):
    pds.config.LIN_REG_EXPR_F64 = dt == torch.float64
    xs = [torch.randn(n, dtype=dt, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * 0.01 * (j % 7 - 3) for j in range(0, p, 5)) + torch.randn(n, dtype=dt, device=dev, generator=gen)
    f = lambda: pds.gram_moments(*xs, target=y, ctx=ctx, out_device=True)
    f(); torch.cuda.synchronize()
    ctx.get_timing(True); ctx.set_timing(True)
    for _ in range(3): f()
    torch.cuda.synchronize(); ctx.set_timing(False)
    t = {k: round(ms / max(c, 1), 3) for k, (ms, c) in ctx.get_timing(True).items() if c}
    gb = n * (p + 1) * xs[0].element_size() / 1e9
    print(f"n={n} p={p} {str(dt)[6:]}: gram {t['moments']} ms  frame {gb:.1f} GB -> {gb / t['moments']:.2f} TB/s")
    del xs, y; torch.cuda.empty_cache()
pds.config.LIN_REG_EXPR_F64 = True
