"""Development aid: per-phase shader-clock sums of the fused grouped kernel (build with EXTRA=-DPDS_PROFILE_PHASES)."""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import _lib
G, R, P = 1_000_000, int(os.environ.get("R", "100")), int(os.environ.get("P", "16"))
N = G * R
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(P)]
y = sum(x * 0.1 for x in xs) + 0.1 * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
so = _lib.load()
buf = (C.c_ulonglong * 8)()
f = lambda: pds.lin_reg_by(*xs, target=y, group_offsets=off, ctx=ctx)
for _ in range(2): f()
torch.cuda.synchronize(); so.pds_debug_phase_cycles(buf, 1)
K = 5
for _ in range(K): f()
torch.cuda.synchronize(); so.pds_debug_phase_cycles(buf, 1)
names = ["tile store (+vmcnt wait)", "next-tile load issue", "flush (acc->LDS->regs)", "solve", "consume (MFMA)", "-", "-", "wave total"]
waves = 2048
tot = buf[7] / K / waves
print(f"per wave: total {tot:.0f} clk (memtime ticks); groups per wave {G / waves:.0f}; ticks per group {tot / (G / waves):.0f}")
for k, n in enumerate(names):
    if n != "-": print(f"  {n:28s} {buf[k] / K / waves:12.0f}  {100.0 * buf[k] / buf[7]:5.1f} %   per group {buf[k] / K / G:8.1f}")
print(f"  {'unaccounted':28s} {(buf[7] - sum(buf[:5])) / K / waves:12.0f}  {100.0 * (buf[7] - sum(buf[:5])) / buf[7]:5.1f} %")
