"""lin_reg_report with a weight column and HC2 / HC3 standard errors at C2's shape (1e8 x 16 f64 + intercept): pass2_kernel<T, true, 2|3, 16>
(residuals + leverages; (X'X)^-1 from the block's LDS copy since round 6) followed by the weighted Gram build of the meat.  Wall ms per call."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
n, P = 100_000_000, 16
xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(P)]
y = sum(xs[j] * ((-1) ** j * (0.05 + 0.03 * j)) for j in range(P)) + 1e-2 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
w = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) + 0.25
for p in (16, 12, 8):
    out = []
    for se in ("se", "hc1", "hc3"):
        f = lambda: pds.lin_reg_report(*xs[:p], target=y, add_bias=True, weights=w, std_err=se, ctx=ctx)
        r = f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        key = {"se": "std_err"}.get(se, f"{se}_se")
        v0 = float((r[key] if key in r else r["std_err"])[0])
        out.append(f"{se} {ms:.2f} ms (se[0] {v0:.9e})")
    print(f"weighted, p = {p:2d}: " + "  ".join(out), flush=True)
