"""Ordered int64 keys against precomputed group offsets on the headline frame (1e6 groups x 100 rows, 16 and 8 f64 features, device
resident): wall ms of pds.lin_reg_by_key and pds.lin_reg_by, their ratio, and the results compared (run once per library variant by
tools/ab_variants.sh; VERDICT r4 item 3: sorted keys <= 1.10 x the offsets wall)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import polars_ds_extension_amd as pds  # noqa: E402
import synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream())
G, R, P = 1_000_000, 100, 16
xs, y = synth.headline_frame(G, R, P, seed=1234)
off = torch.arange(0, G * R + 1, R, dtype=torch.int64, device=dev)
key = torch.arange(7, 7 + G, dtype=torch.int64, device=dev).repeat_interleave(R)


def wall_pair(fa, fb, reps=15):
    """The two calls alternate (the clocks of a box drift over the first seconds of a process: timing one after the other
    compares a cold kernel with a warm one); medians and minima of each."""
    for _ in range(3):
        fa()
        fb()
    ta, tb = [], []
    for _ in range(reps):
        for f, ts in ((fa, ta), (fb, tb)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    ta.sort()
    tb.sort()
    return ta[len(ta) // 2], ta[0], tb[len(tb) // 2], tb[0]


for p in (16, 8):
    t_off, t_off_min, t_key, t_key_min = wall_pair(lambda: pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx),
                                                   lambda: pds.lin_reg_by_key(*xs[:p], target=y, key=key, max_groups=G, ctx=ctx))
    co0, nu0 = pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx)
    k1, co1, nu1 = pds.lin_reg_by_key(*xs[:p], target=y, key=key, max_groups=G, ctx=ctx)
    same = bool(torch.equal(co0, co1) and torch.equal(nu0, nu1) and torch.equal(k1, torch.arange(7, 7 + G, dtype=torch.int64, device=dev)))
    print(f"p = {p:2d}: offsets {t_off:.3f} (min {t_off_min:.3f}) ms, ordered keys {t_key:.3f} (min {t_key_min:.3f}) ms, "
          f"ratio {t_key / t_off:.3f} (of minima {t_key_min / t_off_min:.3f}), keys cost {1e3 * (t_key - t_off):.0f} us, identical results {same}", flush=True)
