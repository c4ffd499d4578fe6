"""Development aid: time the grouped path variants (fused / unfused, qr / choleskey) on the bench frame."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
G, R, P = 1_000_000, 100, int(os.environ.get("P", "16"))
N = G * R
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(P)]
y = sum(x * 0.1 for x in xs) + 0.1 * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
ref = None
for solver in ("qr", "choleskey"):
    for bias in (False,):
        f = lambda: pds.lin_reg_by(*xs, target=y, group_offsets=off, add_bias=bias, solver=solver, ctx=ctx)
        for _ in range(2): f()
        ctx.get_timing(True); ctx.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): co, nu = f()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
        ctx.set_timing(False)
        t = {k: round(ms / 5, 3) for k, (ms, c) in ctx.get_timing(True).items() if c}
        if ref is None: ref = co.clone()
        err = float(((co - ref).norm(dim=1) / ref.norm(dim=1)).max())
        print(f"unfused={os.environ.get('PDS_GROUPED_UNFUSED','0')} solver={solver} bias={bias}: wall {wall*1e3:.3f} ms/step  kernels(ms/step) {t}  max rel diff vs first {err:.2e}")
