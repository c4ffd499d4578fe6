"""Wall time vs kernel time of the fused grouped path, with (solver qr) and without (solver choleskey) the pivoted-QR second pass."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402

G, R = 1_000_000, 100
N = G * R
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(5)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(16)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
import os
_only = os.environ.get("PDS_PROBE_ONLY")
for p, solver in [(8, "choleskey"), (8, "qr"), (8, "choleskey"), (8, "qr")] if _only else [(8, "choleskey"), (8, "qr"), (8, "choleskey"), (8, "qr"), (16, "choleskey"), (16, "qr"), (4, "choleskey"), (4, "qr"),
                  (8, "qr"), (8, "choleskey")]:
    if True:
        for _ in range(3):
            pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, solver=solver, ctx=ctx)
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            co, nu = pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, solver=solver, ctx=ctx)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        ctx.set_timing(False)
        t = ctx.get_timing(reset=True)
        print(f"p={p} solver={solver}: wall {wall:.3f} ms, kernel classes {{k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in t.items() if v[1]}}"
              .replace("{k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in t.items() if v[1]}", str({k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in t.items() if v[1]})),
              f"nulls {int(nu.sum())}", flush=True)
