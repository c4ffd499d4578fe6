"""Timing experiments on the fused 17 .. 32-feature grouped kernel (development build, EXTRA=-DPDS_DEV_SWITCHES): PDS_GMID_DEBUG=1 skips
the matrix steps, =2 skips the hand-over and the solves, =4 hands finished groups over but never solves them (wrong results either way).  Kernel-class times through the library's hooks."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
dev = torch.device("cuda", 0)
G, R = 1_000_000, 100
N = G * R
off = np.arange(0, N + 1, R, dtype=np.int64)
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=g) for _ in range(32)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream())
os.environ["PDS_GMID_VERBOSE"] = "1"
for p in (17, 32):
    for dbg in ("0", "4", "2", "10", "3"):
        os.environ["PDS_GMID_DEBUG"] = dbg
        pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx)
        os.environ.pop("PDS_GMID_VERBOSE", None)
        ctx.get_timing(reset=True); ctx.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
        ctx.set_timing(False)
        tm = ctx.get_timing(reset=True)
        print(f"p={p} PDS_GMID_DEBUG={dbg}: {t * 1e3:.2f} ms wall; kernel classes (ms total, launches): " + ", ".join(f"{k}={v[0]:.2f}/{v[1]}" for k, v in tm.items() if v[1]), flush=True)
