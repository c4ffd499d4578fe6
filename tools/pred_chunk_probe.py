"""Does the 256 MiB memory-side cache keep a chunk of the frame between the fit and the prediction pass?  lin_reg_by_pred over the headline
frame (1e6 groups x 100 rows x 16 f64) as ONE call and as a loop over row chunks of ~CH MB (fit of chunk k, then pred of chunk k, same
stream): summed KERNEL time per kind from the library's HIP-event hooks (host gaps and the per-call synchronisations excluded -- this asks
what the kernels would take if the chunk loop were built into the library)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent))
import synth  # noqa: E402
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
G, R, P = 1_000_000, 100, 16
xs, y = synth.headline_frame(G, R, P, seed=1234, device=dev)
off = torch.arange(0, G * R + 1, R, dtype=torch.int64, device=dev)
def run(chunk_groups):
    def once():
        for g0 in range(0, G, chunk_groups):
            g1 = min(G, g0 + chunk_groups)
            r0, r1 = g0 * R, g1 * R
            pds.lin_reg_by_pred(*[x[r0:r1] for x in xs], target=y[r0:r1], group_offsets=off[g0:g1 + 1] - r0, ctx=ctx)
    once()
    ctx.get_timing(True); ctx.set_timing(True)
    for _ in range(3): once()
    ctx.set_timing(False)
    t = ctx.get_timing(True)
    return {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in t.items() if v[1]}
print("one call        :", run(G), flush=True)
for mb in (400, 200, 150, 100, 50):
    cg = int(mb * 1e6 / ((P + 1) * 8) / R)
    print(f"chunks of {mb:3d} MB :", run(cg), flush=True)
