"""lin_reg_report beyond 16 features (2e7 x 64 f64, 2e7 x 32 f64): wall per std_err type (kernel breakdown: run under rocprofv3 --stats)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
dev = torch.device('cuda', 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
for n, p in ((20_000_000, 64), (20_000_000, 32), (20_000_000, 20)):
    xs = [torch.randn(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * 0.01 * (j % 7 - 3) for j in range(0, p, 5)) + torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    gb = n * (p + 1) * 8 / 1e9
    for name, fn in (("gram", lambda: pds.gram_moments(*xs, target=y, ctx=ctx)), ("lin_reg", lambda: pds.lin_reg(*xs, target=y, add_bias=True, ctx=ctx)),
                     ("report se", lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err="se", ctx=ctx)),
                     ("report hc1", lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err="hc1", ctx=ctx)),
                     ("report hc3", lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err="hc3", ctx=ctx))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"n={n:.0e} p={p} {name:11s}: {ms:7.2f} ms  (frame {gb:.2f} GB: one stream at {gb / ms:.2f} TB/s)", flush=True)
    del xs, y; torch.cuda.empty_cache()
