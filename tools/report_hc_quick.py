"""lin_reg_report at C2 (1e8 x p f64 + bias) for every standard-error type: wall ms and the fraction of the HBM peak of its two streams
(p = 16, 12, 9: the widths whose HC2 / HC3 leverages run on the matrix cores; p = 8: the packed kernels' vector form).
Run once per library variant by tools/ab_variants.sh."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402

dev = torch.device("cuda", 0)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev)
gen.manual_seed(1)
n, P = 100_000_000, 16
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(P)]
y = sum(xs[j] * ((-1) ** j * (0.05 + 0.03 * j)) for j in range(P)) + 1e-2 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
for p in (16, 12, 9, 8):
    out = []
    for se in ("se", "hc1", "hc2", "hc3"):
        f = lambda: pds.lin_reg_report(*xs[:p], target=y, add_bias=True, std_err=se, ctx=ctx)
        r = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        key = {"se": "std_err"}.get(se, f"{se}_se")
        out.append(f"{se} {ms:.2f} ms = {2 * n * (p + 1) * 8 / ms / 1e6 / 8000:.3f} (se[0] {float(r[key][0]):.9e})")
    print(f"p = {p:2d}: " + "  ".join(out), flush=True)
