"""The headline shape (1e6 groups x 100 rows) at 16 .. 32 f64 features: wall ms of lin_reg_by per width (the 16 -> 17 cliff).
PDS_GROUPED_MID_FUSED=0 in the environment: the record pipeline of round 3 (A/B)."""
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
print(f"# PDS_GROUPED_MID_FUSED={os.environ.get('PDS_GROUPED_MID_FUSED', '(unset: fused)')}")
t16 = None
for p in (16, 17, 20, 24, 28, 32):
    pds.lin_reg_by(*xs[:p], target=y, group_offsets=off)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pds.lin_reg_by(*xs[:p], target=y, group_offsets=off)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    t16 = t16 or t
    print(f"1e6 groups x 100 rows x {p} f64: {t * 1e3:.2f} ms ({N * (p + 1) * 8 / t / 1e12:.2f} TB/s algorithmic = {N * (p + 1) * 8 / t / 8e12:.3f} of HBM peak; {t / t16:.2f} x the 16-feature time)", flush=True)
