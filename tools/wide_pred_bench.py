import sys, time
sys.path.insert(0, '.')
import torch
import polars_ds_extension_amd as pds
pds.config.LIN_REG_EXPR_F64 = False
dev = torch.device('cuda', 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
for n, p, dt in ((10_000_000, 512, torch.float32), (20_000_000, 64, torch.float64)):
    pds.config.LIN_REG_EXPR_F64 = dt == torch.float64
    xs = [torch.randn(n, dtype=dt, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * 0.01 * (j % 7 - 3) for j in range(0, p, 5)) + torch.randn(n, dtype=dt, device=dev, generator=gen)
    f = lambda: pds.lin_reg(*xs, target=y, add_bias=True, return_pred=True, ctx=ctx)
    f(); torch.cuda.synchronize()
    ctx.get_timing(True); ctx.set_timing(True)
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ctx.set_timing(False)
    t = {k: round(ms / max(c, 1), 3) for k, (ms, c) in ctx.get_timing(True).items() if c}
    gb = n * (p + 1) * xs[0].element_size() / 1e9
    print(f"n={n} p={p} {dt}: wall {wall*1e3:.1f} ms  kinds {t}  frame {gb:.1f} GB -> pass2 at {gb / t.get('pass2', 1) :.2f} TB/s")
    r = pds.lin_reg_report(*xs, target=y, add_bias=True, std_err="hc1", ctx=ctx) if p <= 64 else None
    if r is not None:
        torch.cuda.synchronize(); t0 = time.perf_counter(); pds.lin_reg_report(*xs, target=y, add_bias=True, std_err="hc3", ctx=ctx); torch.cuda.synchronize()
        print(f"   report hc3 wall {(time.perf_counter()-t0)*1e3:.1f} ms")
    del xs, y; torch.cuda.empty_cache()
