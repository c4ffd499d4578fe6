"""One-box A/B figures of the kernels the round works on (run once per library variant by tools/ab_variants.sh):
   python tools/ab_quick.py [rolling] [grouped] [keyed] [pred]
rolling: C4 rolling / expanding kernel ms;  grouped: fused kernel ms at 1e6 x 100 x 16 and x 8;  keyed: C3 shuffled-keys wall ms;
pred: grouped pred pass (sorted offsets / shuffled keys) wall ms."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import polars_ds_extension_amd as pds  # noqa: E402
import synth  # noqa: E402

what = set(sys.argv[1:]) or {"rolling", "grouped"}
dev = torch.device("cuda", 0)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream())


def kernel_ms(fn, kind, reps=5):
    fn()
    ctx.get_timing(reset=True)
    ctx.get_timing_samples(kind, reset=True)
    ctx.set_timing(True)
    for _ in range(reps):
        fn()
    ctx.set_timing(False)
    s = sorted(ctx.get_timing_samples(kind, reset=True))
    ctx.get_timing(reset=True)
    return s[0], s[len(s) // 2], s[-1], len(s) // reps


def wall_ms(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if "rolling" in what:
    fr = synth.c4_frame(100_000_000, 8, seed=3)
    xs, y = fr["xs"], fr["y"]
    for name, fn in (("rolling w=256", lambda: pds.rolling_lin_reg(*xs, target=y, window_size=256, ctx=ctx)),
                     ("rolling w=252", lambda: pds.rolling_lin_reg(*xs, target=y, window_size=252, ctx=ctx)),
                     ("expanding", lambda: pds.recursive_lin_reg(*xs, target=y, start_with=256, ctx=ctx))):
        lo, med, hi, per = kernel_ms(fn, "rolling", reps=4)
        print(f"{name:16s} 1e8 x 8 f64: kernel ms min/med/max {lo:.3f} {med:.3f} {hi:.3f} ({per} timed brackets per call)", flush=True)
    del xs, y, fr
    torch.cuda.empty_cache()
if "grouped" in what or "pred" in what:
    G, R, P = 1_000_000, 100, 16
    xs, y = synth.headline_frame(G, R, P, seed=1234)
    off = torch.arange(0, G * R + 1, R, dtype=torch.int64, device=dev)
    if "grouped" in what:
        for p in (16, 8):
            lo, med, hi, _ = kernel_ms(lambda: pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx), "grouped_moments", reps=10)
            print(f"grouped fused 1e6 x 100 x {p:2d} f64: kernel ms min/med/max {lo:.4f} {med:.4f} {hi:.4f}", flush=True)
    if "pred" in what:
        for p in (16, 8):
            t_fit = wall_ms(lambda: pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, ctx=ctx))
            t_pred = wall_ms(lambda: pds.lin_reg_by_pred(*xs[:p], target=y, group_offsets=off, ctx=ctx))
            lo, med, hi, _ = kernel_ms(lambda: pds.lin_reg_by_pred(*xs[:p], target=y, group_offsets=off, ctx=ctx), "pass2", reps=5)
            print(f"grouped pred  1e6 x 100 x {p:2d} f64: fit {t_fit:.3f} ms, fit + pred {t_pred:.3f} ms, pred kernel min/med/max {lo:.3f} {med:.3f} {hi:.3f}", flush=True)
    del xs, y
    torch.cuda.empty_cache()
if "keyed" in what or "pred" in what:
    fr = synth.c3_frame(1_000_000, 8, seed=2)
    perm = torch.randperm(fr["n_rows"], device=dev, generator=torch.Generator(device=dev).manual_seed(22))
    ks = fr["keys"][perm]
    xs_s = [x[perm] for x in fr["xs"]]
    y_s = fr["y"][perm]
    del perm
    if "keyed" in what:
        t_sorted = wall_ms(lambda: pds.lin_reg_by(*fr["xs"], target=fr["y"], group_offsets=fr["offsets"], ctx=ctx))
        t_shuf = wall_ms(lambda: pds.lin_reg_by_key(*xs_s, target=y_s, key=ks, ctx=ctx, max_groups=1_000_000))
        print(f"keyed C3: sorted offsets {t_sorted:.3f} ms, shuffled keys {t_shuf:.3f} ms ({t_shuf / t_sorted:.2f}x)", flush=True)
    if "pred" in what:
        t_pk = wall_ms(lambda: pds.lin_reg_by_key_pred(*xs_s, target=y_s, key=ks, ctx=ctx))
        print(f"keyed pred C3 shuffled: fit + pred in frame order {t_pk:.3f} ms", flush=True)
