"""Grouped fits with 17 .. 64 features (one wave per group: the coverage path) -- how far from the stream rate?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
dev = torch.device("cuda", 0)
for G, R, P in ((20_000, 1000, 32), (200_000, 100, 32), (10_000, 1000, 64), (100_000, 100, 64), (200_000, 100, 20)):
    N = G * R
    g = torch.Generator(device=dev); g.manual_seed(1)
    xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=g) for _ in range(P)]
    y = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    off = np.arange(0, N + 1, R, dtype=np.int64)
    pds.lin_reg_by(*xs, target=y, group_offsets=off)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pds.lin_reg_by(*xs, target=y, group_offsets=off)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    gb = N * (P + 1) * 8 / 1e9
    print(f"{G} groups x {R} rows x {P} f64: {t * 1e3:.2f} ms  ({gb:.2f} GB: {gb / t / 1e3:.2f} TB/s; single-regression Gram of the same frame runs at ~5 TB/s)", flush=True)
    del xs, y
