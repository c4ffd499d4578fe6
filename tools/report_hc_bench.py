"""Development aid: lin_reg_report at C2 (1e8 x 16 + bias) for every standard-error type, fused residual pass against
PDS_REPORT_NO_FUSE=1 (separate residual pass + weighted Gram build).  Run on the GPU box; the switch is read once per process."""
import os, subprocess, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

def child():
    import torch
    import polars_ds_extension_amd as pds
    dev = torch.device("cuda", 0)
    ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    n, p = 100_000_000, 16
    xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * ((-1) ** j * (0.05 + 0.03 * j)) for j in range(p)) + 1e-2 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    out = []
    for se in ("se", "hc0", "hc1", "hc2", "hc3"):
        f = lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err=se, ctx=ctx)
        r = f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        key = {"se": "std_err"}.get(se, f"{se}_se")
        out.append(f"{se}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms (se[0] {float(r[key][0]):.6e})")
    print("  ".join(out))

if __name__ == "__main__":
    if os.environ.get("_CHILD"):
        child()
    else:
        for nf in ("0", "1"):
            r = subprocess.run([sys.executable, __file__], env=dict(os.environ, _CHILD="1", PDS_REPORT_NO_FUSE=nf), capture_output=True, text=True, timeout=600)
            print(("fused   " if nf == "0" else "separate"), r.stdout.strip() or r.stderr[-1500:], flush=True)
