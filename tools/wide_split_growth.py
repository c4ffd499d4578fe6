"""The f32 wide Gram's distance to an f64 Gram of the same f32 data against the frame length, both arithmetics (bf16 x 3 split =
default; context option "wide_f32_native" = v_mfma_f32_32x32x2_f32, set per run through Context.set_option): does the split's error stay bounded?
The reference is summed over 1e6-row chunks (torch f64 matmul), so the frame never exists in f64.
   python tools/wide_split_growth.py [p=512] [n=1e6,1e7,3e7] [offset=3.0]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds

p = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("p=")), "512"))
ns = [int(float(v)) for v in next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("n=")), "1e6,1e7,3e7").split(",")]
off = float(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("offset=")), "3.0"))  # mean of every 7th column (0: the C5 frame)
pds.config.LIN_REG_EXPR_F64 = False
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
for n in ns:
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    xs, prev = [], None
    for j in range(p):
        e = torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
        prev = e if prev is None else 0.5 * prev + (0.75 ** 0.5) * e
        xs.append(prev + (off if j % 7 == 0 else 0.0))
    y = sum(xs[j] * (0.1 * (j % 5 - 2)) for j in range(0, p, 16)) + 0.5 * torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
    T = torch.zeros((p + 2, p + 2), dtype=torch.float64, device=dev)
    ck = 1_000_000
    for a in range(0, n, ck):
        Z = torch.stack([c[a:a + ck] for c in xs] + [torch.ones_like(y[a:a + ck]), y[a:a + ck]], dim=1).double()
        T += Z.T @ Z
        del Z
    T = T.cpu().numpy()
    sc = np.sqrt(np.outer(np.diag(T), np.diag(T)))
    for native in ("0", "1"):
        ctx.set_option("wide_f32_native", int(native))
        A = np.asarray(pds.gram_moments(*xs, target=y, ctx=ctx)).astype(np.float64)
        d = np.abs(A - T)
        print(f"n={n:.0e} p={p} offset={off} {'native f32 mfma' if native == '1' else 'bf16 x3 split  '}: fro_rel {np.linalg.norm(d) / np.linalg.norm(T):.3e}  "
              f"max |d| / sqrt(G_ii G_jj) {(d / sc).max():.3e}  diag rel {np.max(np.diag(d) / np.diag(T)):.3e}  "
              f"signed mean (A-T)/sc {np.mean((A - T) / sc):.3e}", flush=True)
    ctx.set_option("wide_f32_native", 0)
    del xs, y
    torch.cuda.empty_cache()
