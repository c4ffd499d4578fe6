"""Is the C5 Gram (1e7 x 512 f32, moments_wide_split256_kernel) bound by what it fetches?  The same launch on a frame whose 512 column
pointers all name ONE 40 MB column: every panel read hits the L2 / the memory-side cache, the HBM stream is 40 MB instead of 20.5 GB --
kernel ms beside the real frame's (HIP events of the library's timing hooks).  If the two agree the kernel is not traffic bound and no
re-mapping of tiles to XCDs can move it."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402
pds.config.LIN_REG_EXPR_F64 = False
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
n, p = 10_000_000, 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
xs = [torch.randn(n, dtype=torch.float32, device=dev, generator=gen) for _ in range(p)]
y = torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
def gram_ms(cols, reps=5):
    for _ in range(2): pds.gram_moments(*cols, target=y, ctx=ctx)
    ctx.get_timing(True); ctx.set_timing(True)
    for _ in range(reps): pds.gram_moments(*cols, target=y, ctx=ctx)
    ctx.set_timing(False)
    ms, cnt = ctx.get_timing(True)["moments"]
    return ms / max(cnt, 1)
a = gram_ms(xs)
b = gram_ms([xs[0]] * p)
c = gram_ms([xs[j % 8] for j in range(p)])
print(f"C5 Gram kernel: real frame {a:.3f} ms | all 512 columns alias one 40 MB column {b:.3f} ms | 8 distinct columns {c:.3f} ms", flush=True)
