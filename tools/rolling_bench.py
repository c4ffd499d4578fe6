"""Rolling / expanding kernel time at C4 (1e8 x 8 f64, w = 256) and a few other shapes: ms per launch via the library's timing hooks."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream())
shapes = [(100_000_000, 8, False, 256, "f64"), (100_000_000, 7, True, 256, "f64"), (50_000_000, 4, False, 64, "f64"), (100_000_000, 8, False, 256, "f32")]
if len(sys.argv) > 1 and sys.argv[1] == "c4":
    shapes = shapes[:1]
for n, p, bias, w, dt in shapes:
    pds.config.LIN_REG_EXPR_F64 = dt == "f64"
    tdt = torch.float64 if dt == "f64" else torch.float32
    xs = [torch.rand(n, dtype=tdt, device=dev, generator=g) for _ in range(p)]
    y = sum(xs[j] * (0.1 * (j + 1)) for j in range(p)) + 1e-3 * torch.randn(n, dtype=tdt, device=dev, generator=g)
    for name, fn in (("rolling", lambda: pds.rolling_lin_reg(*xs, target=y, window_size=w, add_bias=bias, ctx=ctx)),
                     ("expanding", lambda: pds.recursive_lin_reg(*xs, target=y, start_with=w, add_bias=bias, ctx=ctx))):
        fn()
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        for _ in range(3):
            fn()
        ctx.set_timing(False)
        t = ctx.get_timing(reset=True)["rolling"]
        ms = t[0] / t[1]
        pp = p + int(bias)
        es = 8 if dt == "f64" else 4
        alg = n * ((p + 1) * es + (pp + 1) * es + 8)
        print(f"{name:9s} n={n:.0e} p={p} bias={bias} w={w} {dt}: {ms:.3f} ms per call  {alg / ms / 1e6:.0f} GB/s algorithmic = {alg / ms / 1e6 / 8000:.3f} of HBM peak", flush=True)
    del xs, y
    torch.cuda.empty_cache()
pds.config.LIN_REG_EXPR_F64 = True
