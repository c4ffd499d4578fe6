"""
Secondary measurements quoted in DESIGN.md (not the driver's bench line): the other BASELINE configs on one
MI355X, inputs resident in HBM, kernel time from the library's HIP-event hooks, plus a bounded CPU sample of
the oracle where it is the reference's own algorithm (rolling: sequential Woodbury, single thread).
Usage: python tools/bench_extra.py [rolling] [report] [single] [host] [en] [keyed]
"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import polars_ds_extension_amd as pds

which = set(sys.argv[1:]) or {"rolling", "report", "single", "host", "en", "keyed", "c1"}
dev = torch.device("cuda", 0)
ctx = pds.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
out = {}

def timed(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    ctx.get_timing(True); ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
    ctx.set_timing(False)
    t = {k: (ms / max(c, 1), c // reps) for k, (ms, c) in ctx.get_timing(True).items() if c}
    return wall, t, r

if "rolling" in which:
    n, p, w = 100_000_000, 8, 256
    xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * (0.1 * (j + 1)) for j in range(p)) + 1e-3 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    wall, t, (co, pr, va) = timed(lambda: pds.rolling_lin_reg(*xs, target=y, window_size=w, ctx=ctx), reps=3, warm=1)
    alg = n * ((p + 1) * 8 + (p + 1) * 8 + 8)  # 152 B/row at p'=8
    ms = t["rolling"][0]
    out["rolling_c4"] = {"rows": n, "p": p, "window": w, "kernel_ms": round(ms, 3), "wall_ms": round(wall * 1e3, 3),
                         "rows_per_s": round(n / (ms * 1e-3), 1), "algorithmic_GBps": round(alg / (ms * 1e-3) / 1e9, 1)}
    # CPU: the reference's sequential Woodbury chain, 1 thread, bounded sample
    from oracle import oracle as orc
    ns = 2_000_000
    Xh = np.stack([x[:ns].cpu().numpy() for x in xs], axis=1); yh = y[:ns].cpu().numpy()
    t0 = time.perf_counter(); ref = orc.rolling_lr(Xh, yh, w); tc = time.perf_counter() - t0
    err = np.linalg.norm(co[w - 1:ns].cpu().numpy() - ref, axis=1) / np.linalg.norm(ref, axis=1)
    out["rolling_c4"]["cpu_rows_per_s_1thread"] = round(ns / tc, 1)
    out["rolling_c4"]["max_rel_err_vs_chain_2e6_rows"] = float(err.max())
    wall, t, _ = timed(lambda: pds.recursive_lin_reg(*xs, target=y, start_with=16, ctx=ctx), reps=2, warm=1)
    out["recursive_1e8x8"] = {"kernel_ms_total": round(sum(v[0] * v[1] for v in t.values()), 3), "wall_ms": round(wall * 1e3, 3)}
    del xs, y, co, pr, va
    torch.cuda.empty_cache()

if "report" in which or "single" in which or "host" in which:
    n, p = 100_000_000, 16
    xs = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    beta = [(-1.0) ** j * (0.05 + 0.03 * j) for j in range(p)]; beta[3] = 0.0; beta[11] = 0.0
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    for j in range(p): y.add_(xs[j], alpha=beta[j])
    y.add_(torch.randn(n, dtype=torch.float64, device=dev, generator=gen), alpha=1e-2)
    if "single" in which:
        wall, t, b = timed(lambda: pds.lin_reg(*xs, target=y, add_bias=True, ctx=ctx))
        out["single_ols_c2"] = {"wall_ms": round(wall * 1e3, 3), "kernels": {k: round(v[0], 4) for k, v in t.items()},
                                "gram_GBps": round(n * (p + 1) * 8 / (t["moments"][0] * 1e-3) / 1e9, 1)}
    if "report" in which:
        yv = float(y.var(unbiased=True).item())
        for se in ("se", "hc1"):
            wall, t, r = timed(lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err=se, y_var=yv, ctx=ctx), reps=3, warm=1)
            out[f"report_c2_{se}"] = {"wall_ms": round(wall * 1e3, 3), "kernels_ms": {k: [round(v[0], 4), v[1]] for k, v in t.items()},
                                      "p_values": [float(v) for v in r["p>|t|"][[3, 11]]]}
    if "host" in which:
        ns = 20_000_000
        hx = [x[:ns].cpu().numpy() for x in xs]; hy = y[:ns].cpu().numpy()
        pds.lin_reg(*hx, target=hy, add_bias=True, ctx=ctx)
        t0 = time.perf_counter(); pds.lin_reg(*hx, target=hy, add_bias=True, ctx=ctx); th = time.perf_counter() - t0
        out["single_ols_host_buffers"] = {"rows": ns, "wall_s": round(th, 3), "GBps_incl_pcie": round(ns * 17 * 8 / th / 1e9, 2),
                                         "note": "pageable numpy buffers -> hipMemcpyAsync staging -> kernels"}
if "en" in which:
    torch.cuda.empty_cache()
    pds.config.LIN_REG_EXPR_F64 = False
    n, p = 10_000_000, 512
    xs = []
    prev = None
    for j in range(p):
        e = torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
        prev = e if prev is None else 0.5 * prev + (0.75 ** 0.5) * e
        xs.append(prev)
    idx = torch.randperm(p, generator=torch.Generator().manual_seed(4))[:32]
    y = torch.zeros(n, dtype=torch.float32, device=dev)
    cgen = torch.Generator().manual_seed(5)
    for j in idx.tolist(): y.add_(xs[j], alpha=float(torch.randn(1, generator=cgen).item()))
    y.add_(torch.randn(n, dtype=torch.float32, device=dev, generator=gen), alpha=0.5)
    wall, t, b = timed(lambda: pds.lin_reg(*xs, target=y, l1_reg=0.01, l2_reg=0.01, tol=1e-5, ctx=ctx), reps=3, warm=1)
    flops = 2.0 * n * (p + 2) * (p + 2)  # full square, as the reference computes it
    gms = t["moments"][0]
    out["elastic_net_c5"] = {"rows": n, "p": p, "dtype": "f32", "wall_ms": round(wall * 1e3, 2), "gram_ms": round(gms, 3),
                             "gram_TFLOPs_full_square": round(flops / (gms * 1e-3) / 1e12, 1),
                             "gram_TFLOPs_upper_triangle": round(flops / 2 / (gms * 1e-3) / 1e12, 1),
                             "cd_ms": round(t.get("iterative", (0, 0))[0], 3), "nonzero": int((abs(b) > 1e-6).sum()),
                             "arithmetic": "f32 products as three-plane bf16 splits on the bf16 matrix cores (default)"}
    ctx.set_option("wide_f32_native", 1)  # (a context option since round 5: the environment is only its default at pds_ctx_create)
    wall2, t2, _ = timed(lambda: pds.lin_reg(*xs, target=y, l1_reg=0.01, l2_reg=0.01, tol=1e-5, ctx=ctx), reps=3, warm=1)
    ctx.set_option("wide_f32_native", 0)
    out["elastic_net_c5"]["native_f32_mfma"] = {"wall_ms": round(wall2 * 1e3, 2), "gram_ms": round(t2["moments"][0], 3)}
    pds.config.LIN_REG_EXPR_F64 = True
if "c1" in which:
    # configs[0]: pds.lin_reg(x1..x4, target=y, add_bias=False) on a 100k-row f64 frame (benchmarks/test_linear_regression.py:9-31)
    rng = np.random.default_rng(208)
    n = 100_000
    Xh = rng.random((n, 4))
    yh = Xh @ [0.5, 0.25, -0.15, 0.2] + 1e-4 * rng.random(n)
    hc = [np.ascontiguousarray(Xh[:, j]) for j in range(4)]
    dcols = [torch.from_numpy(c).to(dev) for c in hc]
    dy = torch.from_numpy(yh).to(dev)
    res = {}
    for name, f in (("host_columns", lambda: pds.lin_reg(*hc, target=yh, ctx=ctx)), ("device_columns", lambda: pds.lin_reg(*dcols, target=dy, ctx=ctx))):
        for _ in range(5): b = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): b = f()
        torch.cuda.synchronize(); res[name + "_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    res["max_abs_err_vs_numpy"] = float(np.max(np.abs(np.asarray(b.cpu() if hasattr(b, "cpu") else b) - np.linalg.lstsq(Xh, yh, rcond=None)[0])))
    out["config1_100k_x4"] = res

if "keyed" in which:
    # configs[2] with the rows in random order (SURVEY 8d "shuffled variant"): group_by on an unsorted key column
    G, R, p = 1_000_000, 100, 8
    n = G * R
    for a in list(globals()):
        if a in ("xs", "y", "co", "pr", "va"): del globals()[a]
    torch.cuda.empty_cache()
    xs = [torch.randn(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * (0.1 * (j + 1)) for j in range(p)) + 0.1 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    key_sorted = torch.arange(n, dtype=torch.int64, device=dev) // R
    key_shuf = key_sorted[torch.randperm(n, device=dev, generator=gen)]
    res = {}
    for name, key in (("sorted_keys", key_sorted), ("shuffled_rows", key_shuf)):
        torch.cuda.synchronize(); f = lambda: pds.lin_reg_by_key(*xs, target=y, key=key, ctx=ctx)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): k, co, nu = f()
        torch.cuda.synchronize(); res[name] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    out["grouped_by_key_1e6x100x8"] = {"wall_ms": res, "groups": int(k.shape[0]),
                                       "note": "wall time of pds_lr_by_key_f64, frame resident in HBM: order check + run-length encoding "
                                               "(+ radix sort of (key,row) and the gather of 9 columns when the rows are shuffled) + fused grouped kernel"}
print(json.dumps(out, indent=1))
