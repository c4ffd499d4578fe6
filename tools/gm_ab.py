"""A/B aid for the 17 .. 32-feature grouped stream: 1e6 groups x 100 rows, device-resident offsets; kernel ms (HIP events of the library's
timing hooks) and wall ms per width, plus the largest deviation from an f64 torch solve on a sample of groups.   [p=17,24,32] [R=100] [dt=f32]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
G, R = 1_000_000, int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("R=")), "100"))
N = G * R
F32 = "dt=f32" in sys.argv[1:]
if F32: pds.config.LIN_REG_EXPR_F64 = False
DT = torch.float32 if F32 else torch.float64
widths = [int(v) for v in next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("p=")), "17,20,24,28,32").split(",")]
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.randn(N, dtype=DT, device=dev, generator=g) for _ in range(max(widths))]
y = sum(xs[j] * (0.1 * (j % 7 - 3)) for j in range(0, max(widths), 3)) + torch.randn(N, dtype=DT, device=dev, generator=g)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
out = []
for p in widths:
    f = lambda: pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx)
    co, nu = f(); f()
    ctx.get_timing(True); ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    ctx.set_timing(False)
    km = ctx.get_timing(True)["grouped_moments"]
    # parity spot check: 64 groups spread over the frame
    idx = torch.linspace(0, G - 1, 64, device=dev).long()
    err = 0.0
    for gi in idx.tolist():
        X = torch.stack([x[gi * R:(gi + 1) * R] for x in xs[:p]], dim=1).double()
        b = torch.linalg.lstsq(X, y[gi * R:(gi + 1) * R, None].double()).solution[:, 0]
        err = max(err, float(((co[gi].double() - b).norm() / b.norm()).item()))
    out.append(f"p={p}: kernel {km[0] / max(km[1], 1):.3f} ms  wall {wall:.3f} ms  frac {N * (p + 1) * (4 if F32 else 8) / wall / 1e6 / 8000:.3f}  nulls {int(nu.sum())}  max rel err (64 groups) {err:.1e}")
print("\n".join(out), flush=True)
