"""Kernels of an ordered-keys pds_lr_by_key call on the C3-like frame (1e6 groups x 100 rows x 8 f64), device resident."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
dev = torch.device("cuda", 0)
G, R, P = 1_000_000, 100, 8
N = G * R
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=g) for _ in range(P)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
key = torch.arange(G, dtype=torch.int64, device=dev).repeat_interleave(R)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k, c, n = pds.lin_reg_by_key(*xs, target=y, key=key, max_groups=G)
    torch.cuda.synchronize()
    if it >= 2: print(f"ordered keys by_key: {1e3 * (time.perf_counter() - t0):.3f} ms")
