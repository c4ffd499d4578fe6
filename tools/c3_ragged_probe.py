"""Where the C3 frame's sorted-offsets time goes beside the fixed-size frame's: Poisson(100) group sizes vs 100 rows each, with and
without the 0.1 % collinear groups (whose systems go through the marked second pass: pivoted QR + one host round trip).  8 f64 features,
1e6 groups; wall ms of lin_reg_by and the fused kernel's ms (HIP events)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import polars_ds_extension_amd as pds  # noqa: E402
import synth  # noqa: E402
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
for fixed in (None, 100):
    for frac in (1e-3, 0.0):
        fr = synth.c3_frame(1_000_000, 8, seed=2, device=dev, collinear_frac=frac, fixed_size=fixed)
        f = lambda: pds.lin_reg_by(*fr["xs"], target=fr["y"], group_offsets=fr["offsets"], ctx=ctx)
        f(); f()
        ctx.get_timing(True); ctx.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): co, nu = f()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
        ctx.set_timing(False)
        t = ctx.get_timing(True)
        alg = fr["n_rows"] * 9 * 8 + 1_000_000 * (8 * 8 + 17)
        k = {kk: round(v[0] / max(v[1], 1), 3) for kk, v in t.items() if v[1]}
        print(f"sizes {'Poisson(100)' if fixed is None else 'fixed 100  '} collinear {frac:g}: wall {wall:.3f} ms = {alg / wall / 1e6 / 8000:.3f} of HBM; "
              f"kernel ms per launch {k}; nulls {int(nu.sum())}", flush=True)
        del fr
        torch.cuda.empty_cache()
