"""Development aid: the f32 wide Gram on the bf16 matrix cores (three-plane split) against v_mfma_f32_32x32x2_f32
(PDS_WIDE_F32_NATIVE=1) -- accuracy against an f64 Gram of the same f32 data, and time at config 5.
Usage: python tools/wide_split_ab.py [acc] [time] [p=512] [wide=1]  (each arithmetic runs in its own process: the switch is read once)"""
import json, os, subprocess, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

def child(what, p):
    import numpy as np, torch
    import polars_ds_extension_amd as pds
    pds.config.LIN_REG_EXPR_F64 = False
    dev = torch.device("cuda", 0)
    ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    def frame(n, scale):
        xs, prev = [], None
        for j in range(p):
            e = torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
            prev = e if prev is None else 0.5 * prev + (0.75 ** 0.5) * e
            xs.append(prev * scale + (3.0 if j % 7 == 0 else 0.0))   # some columns far from zero mean
        y = sum(xs[j] * (0.1 * (j % 5 - 2)) for j in range(0, p, 37)) + 0.5 * torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
        return xs, y
    out = {}
    if "acc" in what:
        for n, scale in ((1_000_000, 1.0), (300_001, 250.0)):
            xs, y = frame(n, scale)
            A = pds.gram_moments(*xs, target=y, ctx=ctx)
            Z = torch.stack(xs + [torch.ones_like(y), y], dim=1).double()
            T = (Z.T @ Z).cpu().numpy()
            d = np.abs(A.astype(np.float64) - T)
            sc = np.sqrt(np.outer(np.diag(T), np.diag(T)))
            out[f"acc_n{n}_scale{scale}"] = {"max_rel_to_sqrt_diag": float((d / sc).max()), "fro_rel": float(np.linalg.norm(d) / np.linalg.norm(T)),
                                             "max_rel_elementwise_big": float((d / np.abs(T))[np.abs(T) > 1e-3 * sc].max())}
            del Z, xs, y
            torch.cuda.empty_cache()
    if "time" in what:
        n = 10_000_000
        xs, y = frame(n, 1.0)
        for _ in range(2): pds.gram_moments(*xs, target=y, ctx=ctx, out_device=True)
        ctx.get_timing(True); ctx.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): pds.gram_moments(*xs, target=y, ctx=ctx, out_device=True)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
        ctx.set_timing(False)
        t = {k: ms / max(c, 1) for k, (ms, c) in ctx.get_timing(True).items() if c}
        out["time_c5"] = {"wall_ms": round(wall * 1e3, 3), "kernel_ms": {k: round(v, 3) for k, v in t.items()},
                          "useful_TFLOPs_upper_triangle": round(n * (p + 2) * (p + 3) / (wall) / 1e12, 1)}
    print(json.dumps(out))

if __name__ == "__main__":
    if os.environ.get("_WIDE_CHILD"):
        child(set(sys.argv[1:]), int(os.environ.get("_WIDE_P", "512")))
    else:
        what = [a for a in sys.argv[1:] if "=" not in a] or ["acc", "time"]
        p = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("p=")), "512")
        for native in ("0", "1"):
            env = dict(os.environ, _WIDE_CHILD="1", _WIDE_P=p, PDS_WIDE_F32_NATIVE=native)
            r = subprocess.run([sys.executable, __file__] + what, env=env, capture_output=True, text=True, timeout=900)
            print(("native f32 mfma " if native == "1" else "bf16 x3 split   "), r.stdout.strip() or r.stderr[-2000:], flush=True)
