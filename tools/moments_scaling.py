"""Development aid: single-system Gram build time vs number of features (is it HBM-bound at every p?)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
N = 100_000_000
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(16)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
for p in (1, 2, 4, 8, 12, 16):
    for _ in range(2): pds.gram_moments(*xs[:p], target=y, ctx=ctx)
    ctx.get_timing(True); ctx.set_timing(True)
    for _ in range(5): pds.gram_moments(*xs[:p], target=y, ctx=ctx)
    ctx.set_timing(False)
    ms, c = ctx.get_timing(True)["moments"]
    ms /= c
    print(f"p={p:2d}: {ms:.3f} ms  {N*(p+1)*8/ms/1e6:.0f} GB/s")
