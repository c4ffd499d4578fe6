"""Development aid: per-phase shader-clock sums of the paired grouped stream (17 .. 32 features; build with EXTRA=-DPDS_PROFILE_MID)."""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import _lib
G, R = 1_000_000, int(os.environ.get("R", "100"))
N = G * R
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(32)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
off = np.arange(0, N + 1, R, dtype=np.int64)
so = _lib.load()
buf = (C.c_ulonglong * 16)()
names = {0: "stream: waiting for the half-tile", 1: "stream: next half-tile's loads issued", 2: "stream: matrix steps (consume)", 13: "   direct: partial blocks", 12: "   direct: run of whole blocks", 3: "stream: finished group (hand-over, side records)",
         4: "   of it: waiting for the slot", 5: "   publish body (all call sites)", 6: "stream: group advance (offsets)", 11: "stream: per-half-tile look at the stash", 7: "stream wave total", 8: "solver: waiting for a group", 9: "solver: slot -> pending registers", 10: "solver: solve of four", 15: "solver wave total"}
for P in [int(v) for v in os.environ.get("P", "17,24,32").split(",")]:
    f = lambda: pds.lin_reg_by(*xs[:P], target=y, group_offsets=off, ctx=ctx)
    for _ in range(2): f()
    torch.cuda.synchronize(); so.pds_debug_mid_phase_cycles(buf, 1)
    K = 3
    for _ in range(K): f()
    torch.cuda.synchronize(); so.pds_debug_mid_phase_cycles(buf, 1)
    waves = 1024
    print(f"## {P} features, {G} groups x {R} rows: clocks per streaming / solving wave ({G / waves:.0f} groups, {N / 64 / waves:.0f} half-tiles per wave)")
    for k, n in names.items():
        tot = buf[7] if k < 8 else buf[15]
        print(f"  {n:50s} {buf[k] / K / waves:12.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %   per group {buf[k] / K / G:8.1f}   per half-tile {buf[k] / K / (N / 64):8.1f}")
    print(f"  {'stream: group walk and the rest':50s} {(buf[7] - sum(buf[:4])) / K / waves:12.0f}  {100.0 * (buf[7] - sum(buf[:4])) / max(buf[7], 1):5.1f} %")
