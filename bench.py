#!/usr/bin/env python3
"""
bench.py -- headline measurement of the MI355X least-squares path.

A "step" is one pass of the grouped least-squares hot path over one synthetic frame already resident in
HBM: `group_by(key).agg(pds.lin_reg(x1..x16, target=y))` on 1e8 rows x 16 f64 features cut into 1e6
contiguous groups of 100 rows (BASELINE.json: "grouped lstsq regressions/sec ... 1e8 rows x 16 f64
feats").  `value` = regressions/s over the whole job (all ranks), wall clock bracketed by barrier +
synchronize, max over ranks.  The same frame is then fed as ONE regression (config 2, single OLS) to
measure the Gram build GB/s, reported under "gram_build".

  roofline      dominant kernel of the step (the fused per-group Gram + solve kernel): algorithmic bytes per
                launch / average launch duration measured with HIP events on the launch stream (library hooks
                pds_ctx_set_timing / pds_ctx_get_timing), against the 8 TB/s HBM3E peak.
  cpu_baseline  the CPU restatement of the reference (oracle/, kind "port": the Rust crate cannot be
                built here) timed on the host cores of this box on a bounded sample of the same frame
                (p = 16, and p = 8 beside `grouped_p8`).
  end_to_end    the SAME workload when the frame starts in HOST memory as Arrow buffers and goes through the plugin
                boundary (`_polars_plugin_pl_lr_by`, `_polars_plugin_pl_lr`): wall clock, bytes over PCIe, fraction of the
                measured pinned-copy PCIe rate.  This is the rate a Polars user sees; `value` is the HBM-resident rate.
  other_configs   the other BASELINE.json configs on the same box (HBM resident): lin_reg_report at C2 (SE, HC1 and HC3, wall clock of
                the whole call), rolling / expanding fits at C4 (1e8 x 8, window 256; kernel time by HIP events), the elastic net
                of C5 (1e7 x 512 f32; Gram build on the bf16 matrix cores and, for comparison, with the f32 instructions).
  grouped_c3spec  SURVEY.md 8(d)'s C3 data (Poisson(100) sizes in [16, 256], 0.1 % collinear groups -> the rank gate fires,
                8 features), keys sorted and shuffled.

Multi-GPU (--gpus N, launched by torch.distributed.run): STRONG scaling by default -- the 1e6 groups of the fixed 1e8-row
frame are sharded by group key, every rank holds its shard in HBM, and the GATHER of the coefficients and null flags to rank
0 (RCCL point-to-point, piece by piece behind the compute) is inside the timed region.  `--scaling weak` keeps 1e6 groups
per rank (no collective).  `scatter` reports, outside the timed region, what it costs to distribute a frame that is
resident on rank 0 only.  See DESIGN.md "multi-GPU".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PROFILE_TAG = "r06"     # profiles/<tag>_traffic.json: HBM bytes per launch from the rocprofv3 PMC passes


def _gen_frame(torch, dev, seed, G, R, P):
    """x ~ N(0,1), per-group beta ~ N(0,1), noise 0.1; fixed R rows per group (the headline frame; tools/synth.py, shared with
    tests/test_baseline_sizes.py::test_headline_config_against_oracle)."""
    sys.path.insert(0, str(ROOT / "tools"))
    import synth

    return synth.headline_frame(G, R, P, seed=seed, device=dev)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--groups", type=int, default=1_000_000, help="groups of the frame (strong) / per rank (weak)")
    ap.add_argument("--rows-per-group", type=int, default=100)
    ap.add_argument("--feats", type=int, default=16)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--gather-chunks", type=int, default=0, help="pieces per rank: results of a piece travel while the next is computed (0: parallel.auto_chunks)")
    ap.add_argument("--gather", choices=["p2p", "direct"], default="p2p",
                    help="N > 1, strong scaling: how the coefficients reach rank 0 -- RCCL point-to-point per piece (default), or the peers' kernels "
                         "storing straight into rank 0's IPC-mapped result block (opt-in until it has run across two devices)")
    ap.add_argument("--cpu-sample-groups", type=int, default=400_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / grouped_c3spec / scatter (A/B runs)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # PDS_BENCH_DRYRUN=1: a REHEARSAL of the N > 1 control flow on a box without GPUs (tests/test_bench_dryrun.py) -- gloo instead of
    # RCCL, CPU tensors, the per-group compute injected (numpy normal equations: test infrastructure, never a measurement; the line it
    # prints says so in `data`).  Every branch the driver's `--gpus N` run takes -- plan construction, the piece-count agreement, the
    # peer sends, the MAX all-reduce of the elapsed time, the scatter leg -- runs as written; only the kernels are absent.
    dry = os.environ.get("PDS_BENCH_DRYRUN") == "1"
    if not dry:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    # PDS_BENCH_FORCE_DIST=1 runs the RCCL init / barrier / gather code path at world_size 1 (single-GPU smoke of the N > 1 path)
    use_dist = world > 1 or os.environ.get("PDS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import polars_ds_extension_amd as pds
    from polars_ds_extension_amd import parallel as par

    R, P = args.rows_per_group, args.feats
    strong = args.scaling == "strong"
    G_total = args.groups if strong else args.groups * world
    # group-key range partition (fixed group size: balanced in rows == balanced in groups)
    parts = [par.shard_bounds(G_total, world, r) for r in range(world)]
    g_lo, g_hi = parts[rank]
    G = g_hi - g_lo
    N = G * R
    ctx = None
    if not dry:
        ctx = pds.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream(dev))

    def _dry_grouped(xs_, y_, off_, add_bias=False, **kw):  # (rehearsal only: per-group normal equations in numpy)
        o = np.asarray(off_.cpu() if hasattr(off_, "cpu") else off_, dtype=np.int64)
        X = np.stack([np.asarray(x) for x in xs_], axis=1)
        yy = np.asarray(y_)
        co = np.empty((len(o) - 1, X.shape[1]))
        for k in range(len(o) - 1):
            Xg = X[o[k]:o[k + 1]]
            co[k] = np.linalg.solve(Xg.T @ Xg, Xg.T @ yy[o[k]:o[k + 1]])
        return torch.from_numpy(co), torch.zeros(len(o) - 1, dtype=torch.uint8)

    # ---- this rank's shard of the synthetic frame, generated in HBM (seeded per rank)
    xs, y = _gen_frame(torch, dev, 1234 + rank, G, R, P)
    offsets = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
    off_host = np.arange(0, N + 1, R, dtype=np.int64)
    if not dry:
        torch.cuda.synchronize(dev)
    gather = use_dist and strong

    class _Off:  # device offsets for the kernels, host offsets for the piece bounds (no device read-back when the plan is built)
        def __init__(self, d, h):
            self.d, self.h = d, h

        def cpu(self):
            return self.h

        def __getitem__(self, s):
            return self.d[s]

    off_pair = _Off(offsets, torch.from_numpy(off_host))
    # N-rank step: the prepared plan (persistent result buffers, the root's shard fitted in place, one grouped send per piece)
    # --gather direct (opt-in): the peers' kernels store straight into rank 0's result block; default: RCCL point-to-point
    plan = par.GroupedShardPlan(xs, y, off_pair, parts, rank=rank, gather_to=0, chunks=args.gather_chunks or None, ctx=ctx,
                                grouped_fn=_dry_grouped if dry else None, direct=(args.gather == "direct" and not dry),
                                add_bias=False) if gather else None
    chunks = plan.chunks if plan else 1

    def step():
        if gather:
            res = plan.step()
            return (res[2], res[3]) if rank == 0 else (res[0], res[1])
        if dry:
            return _dry_grouped(xs, y, off_host)
        return pds.lin_reg_by(*xs, target=y, group_offsets=offsets, add_bias=False, ctx=ctx)

    def barrier():
        if not dry:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    if not dry:
        ctx.get_timing(reset=True)
        ctx.get_timing_samples("grouped_moments", reset=True)
        ctx.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        coeffs, nulls = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dry:
        timing = {"grouped_moments": (0.0, 0), "solve": (0.0, 0)}
        launch_ms = []
    else:
        ctx.set_timing(False)
        timing = ctx.get_timing(reset=True)
        launch_ms = sorted(ctx.get_timing_samples("grouped_moments", reset=True))
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = G_total / (elapsed / args.steps)

    # ---- roofline of the dominant kernel (fused grouped Gram + solve), rank 0's launches
    q = P + 2
    gm_ms, gm_cnt = timing["grouped_moments"]
    sv_ms, sv_cnt = timing["solve"]
    launches_per_step = max(gm_cnt // max(args.steps, 1), 1)
    groups_per_launch = G / launches_per_step
    fused = sv_cnt == 0  # Gram + solve in one kernel: no moment record traffic
    # algorithmic bytes per launch: every input element once + offsets, plus what the kernel must
    # write: the coefficients (fused) or the (p+2)^2 moment record (two-kernel pipeline)
    bytes_per_group = R * (P + 1) * 8 + 16 + (P * 8 + 1 if fused else q * q * 8)
    alg_bytes = groups_per_launch * bytes_per_group
    avg_ms = gm_ms / max(gm_cnt, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch measured offline with rocprofv3 PMC passes (profiles/<tag>_traffic.json, committed)
    traffic = None
    traffic_source = None
    for tag in (PROFILE_TAG, "r04", "r03", "r02", "r01"):
        try:
            tj = json.loads((ROOT / "profiles" / f"{tag}_traffic.json").read_text())["kernels"]
            want = "pds::grouped_stream_kernel<double, 16," if fused else "pds::grouped_moments_kernel<double>"
            hit = [v for k, v in tj.items() if k.startswith(want)]
            if hit and G == 1_000_000 and R == 100 and P == 16 and launches_per_step == 1:
                traffic = int(hit[0]["hbm_bytes_per_launch"])
                traffic_source = f"profiles/{tag}_traffic.json (offline rocprofv3 PMC passes of this kernel on this workload, not a counter of this run)"
                break
        except Exception:
            continue
    roofline = {
        "bound": "hbm", "kernel": "grouped_stream_kernel<double,16,cholesky> (Gram + solve fused)" if fused else "grouped_moments_kernel<double>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
        "launch_ms_min_median_max": [round(launch_ms[0], 4), round(launch_ms[len(launch_ms) // 2], 4), round(launch_ms[-1], 4)] if launch_ms else None,
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "solve_ms_per_step": round(sv_ms / max(args.steps, 1), 4), "gram_ms_per_step": round(gm_ms / max(args.steps, 1), 4),
    }

    # ---- what this box's HBM delivers to a plain streaming copy (quote the spec peak AND a measured figure)
    if rank == 0 and not dry:
        a = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize(dev)
        copy_gbps = 5 * 2 * (1 << 30) / (time.perf_counter() - t0) / 1e9
        # (read + write bytes of a 1 GiB device-to-device copy: a reference figure of the box, NOT a ceiling for a read-only stream --
        #  the achievable read stream of this box is `read_stream_GBps` below, the single-OLS Gram kernel on the same frame)
        roofline["measured_copy_GBps"] = round(copy_gbps, 1)
        del a, b

    # ---- fixed cost of the N-rank step, measured on this one GPU: the same frame through GroupedShardPlan over an RCCL process
    # group of world size 1 (no peer: what is left is everything a step does beside its kernel -- Python, the prepared C call, the
    # stream bookkeeping) against the plain call timed above.  At N ranks the kernel shrinks by N, this does not.
    dist_overhead = None
    if rank == 0 and world == 1:
        try:
            dist_overhead = _dist_step_overhead(torch, par, pds, ctx, dev, xs, y, off_pair, offsets, args, ms_per_step, use_dist)
        except Exception as e:
            dist_overhead = {"error": f"{type(e).__name__}: {e}"}

    # ---- config 2 on the same frame: single OLS Gram build (pds_moments), HBM GB/s
    gram = None
    if rank == 0 and not dry:
        for _ in range(2):
            pds.gram_moments(*xs, target=y, ctx=ctx)
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        reps = 5
        for _ in range(reps):
            pds.gram_moments(*xs, target=y, ctx=ctx)
        ctx.set_timing(False)
        ms, cnt = ctx.get_timing(reset=True)["moments"]
        gb = N * (P + 1) * 8 / 1e9
        g_ms = ms / max(cnt, 1)
        gram = {"workload": f"single OLS Gram build, {N:.0e} rows x {P} f64 feats", "algorithmic_GB": round(gb, 3),
                "avg_launch_ms": round(g_ms, 4), "achieved_GBps": round(gb / (g_ms * 1e-3), 1),
                "frac_of_hbm_peak": round(gb / (g_ms * 1e-3) / HBM_PEAK_GBPS, 4)}
        roofline["read_stream_GBps"] = gram["achieved_GBps"]  # what a pure read stream of these columns reaches on this box (Gram kernel)
        roofline["frac_of_read_stream"] = round(achieved / gram["achieved_GBps"], 4) if gram["achieved_GBps"] else None

    # ---- BASELINE.json configs[2] as written (8 features; the headline metric is quoted on 16 -- SURVEY.md 8d asks for
    # both): the same groups on the first 8 feature columns
    p8 = None
    if rank == 0 and P >= 8 and not dry:
        for _ in range(2):
            pds.lin_reg_by(*xs[:8], target=y, group_offsets=offsets, add_bias=False, ctx=ctx)
        torch.cuda.synchronize(dev)
        reps = 5
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        t8 = time.perf_counter()
        for _ in range(reps):
            pds.lin_reg_by(*xs[:8], target=y, group_offsets=offsets, add_bias=False, ctx=ctx)
        torch.cuda.synchronize(dev)
        t8 = (time.perf_counter() - t8) / reps
        ctx.set_timing(False)
        k8_ms, k8_cnt = ctx.get_timing(reset=True)["grouped_moments"]
        b8 = G * (R * 9 * 8 + 16 + 8 * 8 + 1)
        k8 = k8_ms / max(k8_cnt, 1)
        p8 = {"workload": f"{G} groups x {R} rows x 8 f64 feats", "regressions_per_s": round(G / t8, 1),
              "ms_per_step": round(t8 * 1e3, 4), "kernel_ms": round(k8, 4), "algorithmic_GBps": round(b8 / (k8 * 1e-3) / 1e9, 1),
              "frac_of_hbm_peak": round(b8 / (k8 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

    # ---- the same step from the KEY COLUMN (sorted int64 keys, what group_by(key) starts from) instead of precomputed offsets: one pass
    # over the keys (order check + run marks), a scan and a pass over the marks in front of the same fused kernel
    by_key = None
    if rank == 0 and not dry:
        try:
            by_key = _by_key(torch, pds, ctx, dev, xs, y, offsets, G, R, P)
        except Exception as e:
            by_key = {"error": f"{type(e).__name__}: {e}"}

    # ---- CPU baseline + parity spot check on a bounded sample (rank 0, N = 1 only)
    cpu = None
    parity = None
    host_cols = None
    if rank == 0 and world == 1 and not args.no_cpu and not dry:
        cpu, parity = _cpu_baseline(np, args, xs, y, coeffs, nulls, G, R, P)

    # ---- the same workload from HOST Arrow buffers through the plugin boundary (rank 0, N = 1): the rate a Polars user sees
    end_to_end = None
    if rank == 0 and world == 1 and not args.no_extras and not dry:
        try:
            end_to_end = _end_to_end(torch, np, pds, dev, xs, y, G, R, P, cpu)
        except Exception as e:  # pyarrow / harness trouble must not cost the headline line
            end_to_end = {"error": f"{type(e).__name__}: {e}"}

    # ---- the other BASELINE configs, on the headline frame where they share it (C2) and on a C4 frame
    other = None
    if rank == 0 and world == 1 and not args.no_extras and not dry:
        try:
            other = _other_configs(torch, pds, ctx, dev, xs, y, N, P, with_cpu=not args.no_cpu)
        except Exception as e:
            other = {"error": f"{type(e).__name__}: {e}"}

    # ---- SURVEY.md 8(d) C3 data: Poisson sizes, collinear groups (the gate fires inside the timed run), sorted and shuffled keys
    c3 = None
    if rank == 0 and world == 1 and not args.no_extras and not dry:
        try:
            del xs, y
            torch.cuda.empty_cache()
            c3 = _c3_spec(torch, pds, ctx, dev)
        except Exception as e:
            c3 = {"error": f"{type(e).__name__}: {e}"}

    # ---- scatter leg (outside the timed region): what distributing a rank-0-resident frame of this size costs
    scatter = None
    if use_dist and strong and not args.no_extras:
        try:
            scatter = _scatter_leg(torch, dist, par, dev, rank, world, G_total, R, P)
        except Exception as e:
            scatter = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        per = "of the frame" if strong else "per GPU"
        line = {
            "metric": "grouped lstsq regressions/sec (1e8 rows x 16 f64 feats; Gram-build GB/s under gram_build)",
            "value": round(value, 1), "unit": "regressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not dry else "synthetic -- DRY RUN on CPU (gloo, injected numpy compute): control-flow rehearsal, not a measurement",
            "config": {"workload": f"group_by(key).agg(lin_reg): {args.groups} groups x {R} rows x {P} f64 feats {per} "
                                   f"({args.groups * R:.0e} rows), OLS with rank gate (pl_lr default path), inputs resident in HBM; `value`: the groups "
                                   f"given as row offsets (pre-segmented, as Polars hands groups to pl_lr -- the cpu_baseline gets the same); "
                                   f"the same step starting from a sorted int64 key column: grouped_by_key"
                                   + (", coefficients + null flags gathered to rank 0 inside the timed region" if gather else ""),
                       "groups_total": G_total, "groups_per_gpu": G, "rows_per_group": R, "features": P,
                       "parallelism": f"group-sharded x{world}", "gather_chunks": chunks if gather else None,
                       "gather": (args.gather if gather else None)},
            "roofline": roofline, "gram_build": gram, "grouped_p8": p8, "grouped_by_key": by_key, "cpu_baseline": cpu, "parity_spot_check": parity,
            "dist_step": dist_overhead, "end_to_end": end_to_end, "other_configs": other, "grouped_c3spec": c3, "scatter": scatter,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _by_key(torch, pds, ctx, dev, xs, y, offsets, G, R, P):
    """lin_reg_by_key on sorted int64 keys against lin_reg_by on the offsets of the same groups, p = P and p = 8: wall ms (medians of 9 alternating calls)."""
    keys = torch.arange(G, dtype=torch.int64, device=dev).repeat_interleave(R)
    out = {"workload": f"{G} groups x {R} rows, sorted int64 key column resident in HBM (0.8 GB per 1e8 rows) instead of group offsets"}

    def wall_pair(fa, fb, reps=9):  # (alternating calls: a box's clocks drift over a process's first seconds)
        for _ in range(2):
            fa()
            fb()
        ta, tb = [], []
        for _ in range(reps):
            for f, ts in ((fa, ta), (fb, tb)):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
        return sorted(ta)[len(ta) // 2], sorted(tb)[len(tb) // 2]

    for q in sorted({P, 8}, reverse=True):
        if q > P:
            continue
        t_off, t_key = wall_pair(lambda: pds.lin_reg_by(*xs[:q], target=y, group_offsets=offsets, add_bias=False, ctx=ctx),
                                 lambda: pds.lin_reg_by_key(*xs[:q], target=y, key=keys, max_groups=G, ctx=ctx))
        alg = G * R * (q + 2) * 8 + G * (q * 8 + 1 + 8)  # features + target + keys in; coefficients + flags + distinct keys out
        out[f"p{q}"] = {"by_key_wall_ms": round(t_key * 1e3, 4), "offsets_wall_ms": round(t_off * 1e3, 4), "ratio": round(t_key / t_off, 4),
                        "regressions_per_s": round(G / t_key, 1), "algorithmic_bytes": int(alg),
                        "frac_of_hbm_peak": round(alg / t_key / 1e9 / HBM_PEAK_GBPS, 4)}
    del keys
    return out


def _dist_step_overhead(torch, par, pds, ctx, dev, xs, y, off_pair, offsets, args, plain_ms, have_pg):
    """World-1 RCCL process group + GroupedShardPlan (gather_to = 0) on the headline shard: wall per step beside the plain call's."""
    import socket

    import torch.distributed as dist

    made = False
    if not have_pg:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        made = True
    try:
        G = len(off_pair.h) - 1
        plan = par.GroupedShardPlan(xs, y, off_pair, [(0, G)], rank=0, gather_to=0, ctx=ctx, add_bias=False)

        def timed(fn, steps):
            for _ in range(args.warmup):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            return 1e3 * (time.perf_counter() - t0) / steps

        steps = max(args.steps, 10)
        # interleaved repeats: (plan, plain) pairs on the same box state; the medians are compared
        pl, pn = [], []
        for _ in range(3):
            pl.append(timed(plan.step, steps))
            pn.append(timed(lambda: pds.lin_reg_by(*xs, target=y, group_offsets=offsets, add_bias=False, ctx=ctx), steps))
        pl_ms, pn_ms = sorted(pl)[1], sorted(pn)[1]
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        for _ in range(steps):
            plan.step()
        ctx.set_timing(False)
        k_ms, k_cnt = ctx.get_timing(reset=True)["grouped_moments"]
        k_ms = k_ms / max(k_cnt, 1)
        return {"what": "GroupedShardPlan.step over an RCCL group of world size 1 (gather_to = 0, no peer) on the headline shard, beside the "
                        "plain lin_reg_by call; barrier + synchronize on both sides, median of 3 interleaved repeats",
                "plan_step_ms": round(pl_ms, 4), "plain_call_ms": round(pn_ms, 4), "kernel_ms": round(k_ms, 4),
                "dist_step_overhead_us": round(1e3 * (pl_ms - pn_ms), 1),
                "step_minus_kernel_us": round(1e3 * (pl_ms - k_ms), 1), "plain_minus_kernel_us": round(1e3 * (pn_ms - k_ms), 1),
                "chunks": plan.chunks}
    finally:
        if made:
            dist.destroy_process_group()


def _cpu_baseline(np, args, xs, y, coeffs, nulls, G, R, P):
    """
    The CPU restatement of the reference beside the GPU figure, on this box's host cores.

    Two builds of the SAME restatement (oracle/): the PERFORMANCE build (oracle/pds_perf.c: -O3 -march=native, FMA and vectorised
    reductions allowed, compiled here, on all hardware threads this process may use, frame pages first touched by the worker
    threads, one warm-up pass, >= 2 s of timed passes) is `value`; the PARITY build (-O2 -march=x86-64-v3 -ffp-contract=off,
    the checker of tests/) is timed beside it for continuity with earlier rounds and provides the parity spot check.  Host
    STREAM triad is printed for context: a Gram build cannot beat it.
    """
    from oracle import oracle as orc

    orc.build()
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gs = min(args.cpu_sample_groups, G)
    ns = gs * R
    host_cols = [y[:ns].cpu().numpy()] + [x[:ns].cpu().numpy() for x in xs]
    off_h = np.arange(0, ns + 1, R, dtype=np.int64)
    cpu = {"unit": "regressions/s", "cores": nthreads, "kind": "port", "nproc": os.cpu_count()}
    # ---- parity build (the checker): one pass, OpenMP over groups -- also the parity spot check of the timed GPU result
    nt_par = min(nthreads, orc.max_threads()) if orc.max_threads() > 0 else nthreads
    orc.grouped_lr(host_cols, off_h[: 1001], nthreads=nt_par)  # warm the thread pool
    t1 = time.perf_counter()
    co_cpu, nu_cpu = orc.grouped_lr(host_cols, off_h, add_bias=False, tol=1e-12, nthreads=nt_par)
    t_par = time.perf_counter() - t1
    co_gpu = coeffs[:gs].cpu().numpy()
    num = np.linalg.norm(co_gpu - co_cpu, axis=1)
    den = np.linalg.norm(co_cpu, axis=1)
    parity = {"groups_checked": int(gs), "max_normwise_rel_err": float(np.max(num / den)),
              "null_mismatches": int(np.sum(nulls[:gs].cpu().numpy().astype(bool) != nu_cpu))}
    t2 = time.perf_counter()
    orc.gram_cols(host_cols, nthreads=nt_par)
    t_gram_par = time.perf_counter() - t2
    cpu["parity_build"] = {"flags": "-O2 -march=x86-64-v3 -ffp-contract=off -fopenmp", "threads": nt_par,
                           "regressions_per_s": round(gs / t_par, 1), "gram_build_GBps": round(ns * (P + 1) * 8 / t_gram_par / 1e9, 2),
                           "note": "the parity checker's build, numpy-allocated frame (one NUMA node), dynamic schedule; earlier rounds' figure"}
    # ---- performance build
    try:
        orc.build_perf()
        frame = orc.PerfFrame(host_cols, nthreads)
        co_p, nu_p, t_one = orc.perf_grouped_lr(frame, off_h, passes=1)
        passes = max(1, int(np.ceil(2.5 / max(t_one, 1e-6))))
        _, _, t_all = orc.perf_grouped_lr(frame, off_h, passes=passes)
        ok = ~nu_cpu
        dist = float(np.max(np.linalg.norm(co_p[ok] - co_cpu[ok], axis=1) / den[ok]))
        cpu["value"] = round(passes * gs / t_all, 1)
        cpu["build"] = "gcc -O3 -march=native -fopenmp -fno-math-errno -fassociative-math -fno-signed-zeros -fno-trapping-math (oracle/pds_perf.c, compiled on this box)"
        cpu["sample"] = (f"{gs} groups x {R} rows x {P} f64 feats (first {ns} rows of the same frame), {passes} timed passes = {t_all:.2f} s "
                         f"after a warm-up pass, {nthreads} OpenMP threads over contiguous group ranges, pages first touched by their "
                         f"worker; per group: marshalling copy + X'X + X'y + gated col-piv QR, as Polars drives pl_lr")
        cpu["perf_vs_parity_build_max_rel"] = dist
        cpu["perf_null_mismatches"] = int(np.sum(nu_p != nu_cpu))
        if P >= 8:
            _, _, t8 = orc.perf_grouped_lr(frame, off_h, n_features=8, passes=passes)
            cpu["p8_value"] = round(passes * gs / t8, 1)
            cpu["p8_sample"] = f"{gs} groups x {R} rows x 8 f64 feats, {nthreads} threads, {passes} passes = {t8:.2f} s"
        _, t_gram = orc.perf_gram_cols(frame, reps=5)
        cpu["gram_build_GBps"] = round(ns * (P + 1) * 8 / t_gram / 1e9, 2)
        cpu["gram_build_sample"] = f"{ns} rows x {P + 1} f64 columns, [X y]'[X y] straight from the column buffers, {nthreads} threads, best of 5 = {t_gram * 1e3:.1f} ms"
        cpu["stream_triad_GBps"] = round(orc.perf_stream_triad(1 << 28, nthreads, 3), 1)
        cpu["stream_triad_note"] = "a[i] = b[i] + s c[i], 3 x 2 GiB, all threads, first touch by the workers, best of 3: the host memory roof of a Gram build"
        frame.close()
    except Exception as e:  # a box without gcc must not cost the headline line: fall back to the parity build's figure, say so
        cpu["value"] = cpu["parity_build"]["regressions_per_s"]
        cpu["perf_build_error"] = f"{type(e).__name__}: {e}"
        cpu["sample"] = f"{gs} groups x {R} rows x {P} f64 feats, parity build only, {t_par:.2f} s"
    # sanity bound: numpy / OpenBLAS X'X of the same sample, as many threads as it takes by default
    try:
        Zs = np.stack(host_cols[1:] + [host_cols[0]], axis=1)  # row-major copy, not timed
        t3 = time.perf_counter()
        Zs.T @ Zs
        t_np = time.perf_counter() - t3
        cpu["gram_build_numpy_GBps"] = round(ns * (P + 1) * 8 / t_np / 1e9, 2)
        del Zs
    except Exception:
        pass
    return cpu, parity


def _end_to_end(torch, np, pds, dev, xs, y, G, R, P, cpu):
    """Host Arrow buffers -> `_polars_plugin_pl_lr_by` / `_polars_plugin_pl_lr` -> Arrow result; wall clock."""
    import ctypes as C

    import pyarrow as pa

    sys.path.insert(0, str(ROOT / "tests"))
    import plugin_harness as ph  # pyarrow + ctypes in the role of the Polars engine (test infrastructure)
    from polars_ds_extension_amd import _lib

    lib = _lib.load()
    N = G * R
    # measured PCIe rate of this box: pinned host -> HBM, 1 GiB
    pin = torch.empty(1 << 27, dtype=torch.float64).pin_memory()
    dst = torch.empty(1 << 27, dtype=torch.float64, device=dev)
    dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(dev)
    pcie = 3 * (1 << 30) / (time.perf_counter() - t0) / 1e9
    del pin, dst
    host = [("y", pa.array(y.cpu().numpy()))] + [(f"x{j + 1}", pa.array(xs[j].cpu().numpy())) for j in range(P)]
    key = ("key", pa.array(np.repeat(np.arange(G, dtype=np.int64), R)))
    kw = {"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
          "weighted": False, "positive": False, "singular_x_tol": 1e-12}
    out = {"pcie_pinned_h2d_GBps": round(pcie, 1)}

    def timed(sym, ins, reps=3, warm=1):
        for _ in range(warm):  # (large results: the second call still allocates -- the first result's pinned blocks are alive)
            _, res = ph.call_plugin(lib, sym, ins, kw)
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            _, res = ph.call_plugin(lib, sym, ins, kw)
            ts.append(time.perf_counter() - t1)
        return float(np.median(ts)), res

    # default route: the frame in row slices through two contexts on this device (a slice crosses PCIe while the previous one is
    # fitted and copied back; with PDS_DEVICES=0,1,.. every device pulls its own slices over its own link), results into pinned
    # Arrow buffers.  Beside it the single-context route of the earlier rounds (PDS_BY_KEY_MULTI_MIN_ROWS=0).
    t_by, res = timed("pl_lr_by", [key] + host)
    assert len(res) == G
    gb_by = (N * (P + 2) * 8 + G * (P * 8 + 8 + 1)) / 1e9  # keys + y + features up, keys + coefficients + validity down
    out["pl_lr_by"] = {"workload": f"host Arrow frame, {G} groups x {R} rows x {P} f64 feats + int64 keys", "wall_ms": round(t_by * 1e3, 1),
                       "regressions_per_s": round(G / t_by, 1), "pcie_GB": round(gb_by, 2), "GBps": round(gb_by / t_by, 1),
                       "frac_of_pcie_rate": round(gb_by / t_by / pcie, 3),
                       "route": f"sliced, {os.environ.get('PDS_BY_KEY_CONTEXTS', '2')} contexts per device, devices {os.environ.get('PDS_DEVICES', '0')}"}
    os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"] = "0"
    lib.pds_plugin_reload_settings()  # (the plugin layer reads its environment once)
    try:
        t_by1, _ = timed("pl_lr_by", [key] + host, reps=2)
    finally:
        del os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"]
        lib.pds_plugin_reload_settings()
    out["pl_lr_by"]["single_context_wall_ms"] = round(t_by1 * 1e3, 1)
    out["pl_lr_by"]["single_context_frac_of_pcie_rate"] = round(gb_by / t_by1 / pcie, 3)
    # `.over(key)` / group_by().agg(lin_reg(return_pred=True)) on the same host frame: pred + resid of every row come back
    # (PCIe is full duplex: with slices the predictions of slice s travel down while slice s + 1 travels up)
    t_bp, resp = timed("pl_lr_by_pred", [key] + host, reps=3, warm=2)
    assert len(resp) == N
    os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"] = "0"
    lib.pds_plugin_reload_settings()
    try:
        t_bp1, _ = timed("pl_lr_by_pred", [key] + host, reps=3, warm=1)
    finally:
        del os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"]
        lib.pds_plugin_reload_settings()
    del resp
    out["pl_lr_by_pred"] = {"workload": "same host frame, per-row pred + resid of every group's fit (Struct{pred,resid}, frame order)",
                            "wall_ms": round(t_bp * 1e3, 1), "up_GB": round(N * (P + 2) * 8 / 1e9, 2), "down_GB": round(N * 17 / 1e9, 2),
                            "frac_of_pcie_rate_up": round(N * (P + 2) * 8 / 1e9 / t_bp / pcie, 3), "single_context_wall_ms": round(t_bp1 * 1e3, 1)}
    t_lr, _ = timed("pl_lr", host)
    gb_lr = N * (P + 1) * 8 / 1e9
    out["pl_lr"] = {"workload": f"host Arrow frame, single OLS {N:.0e} rows x {P} f64 feats", "wall_ms": round(t_lr * 1e3, 1),
                    "pcie_GB": round(gb_lr, 2), "GBps": round(gb_lr / t_lr, 1), "frac_of_pcie_rate": round(gb_lr / t_lr / pcie, 3)}
    if cpu:
        out["gpu_over_cpu_host_resident"] = round(G / t_by / cpu["value"], 1)
        out["note"] = ("host-resident frames are PCIe bound: this ratio, not the HBM-resident one, is what an unchanged Polars "
                       "query sees on one GPU")
    return out


def _mid_width(torch, np, pds, ctx, dev, wall):
    """17 .. 64 features (between the two BASELINE widths): single-regression Gram / robust report on 2e7 x 32 f64, and the grouped fit
    of 200 000 groups x 100 rows x 32 -- the streaming multi-tile-column kernels and the wave-per-system solver."""
    n, p = 20_000_000, 32
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    xs = [torch.randn(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    y = sum(xs[j] * (0.05 * (j + 1)) for j in range(0, p, 5)) + torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    gb = n * (p + 1) * 8 / 1e9
    out = {"workload": f"{n:.0e} rows x {p} f64 features"}
    ms = wall(lambda: pds.gram_moments(*xs, target=y, ctx=ctx))
    out["gram_ms"] = round(ms, 3)
    out["gram_frac_of_hbm_peak"] = round(gb / ms * 1e3 / HBM_PEAK_GBPS, 4)
    for se in ("se", "hc3"):
        out[f"report_{se}_ms"] = round(wall(lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err=se, ctx=ctx)), 3)
    G, R = 200_000, 100
    off = np.arange(0, G * R + 1, R, dtype=np.int64)
    ms = wall(lambda: pds.lin_reg_by(*xs, target=y, group_offsets=off, ctx=ctx))
    out["grouped_200000x100_ms"] = round(ms, 3)
    out["grouped_regressions_per_s"] = round(G / ms * 1e3, 1)
    # the 16 -> 17 feature step under group_by: the same 2e5 groups x 100 rows at 16 / 17 / 24 / 32 features, and the
    # 64-feature frame whose random groups sit at the reference's rank gate (1e5 groups x 100 rows: the first half of the columns twice)
    steps = {}
    for q in (16, 17, 24, 32):
        steps[str(q)] = round(wall(lambda: pds.lin_reg_by(*xs[:q], target=y, group_offsets=off, ctx=ctx)), 3)
    out["grouped_200000x100_by_features_ms"] = steps
    out["grouped_ratio_17_over_16"] = round(steps["17"] / steps["16"], 3)
    out["grouped_ratio_32_over_16"] = round(steps["32"] / steps["16"], 3)
    G2 = 100_000
    off2 = np.arange(0, G2 * R + 1, R, dtype=np.int64)
    xs2 = [x[: G2 * R] for x in xs] + [torch.randn(G2 * R, dtype=torch.float64, device=dev, generator=gen) for _ in range(32)]
    co, nu = pds.lin_reg_by(*xs2, target=y[: G2 * R], group_offsets=off2, ctx=ctx)
    out["grouped_100000x100x64_ms"] = round(wall(lambda: pds.lin_reg_by(*xs2, target=y[: G2 * R], group_offsets=off2, ctx=ctx)), 3)
    out["grouped_100000x100x64_null_groups"] = int(nu.sum())
    del xs, y, xs2
    torch.cuda.empty_cache()
    return out


def _other_configs(torch, pds, ctx, dev, xs, y, N, P, with_cpu=True):
    out = {}

    def wall(fn, reps=3):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3

    gb = N * (P + 1) * 8 / 1e9
    for se in ("se", "hc1", "hc3"):
        ms = wall(lambda: pds.lin_reg_report(*xs, target=y, add_bias=True, std_err=se, ctx=ctx))
        streams = 2  # Gram + one more pass (residual pass, or the fused residual + meat pass)
        out[f"report_c2_{se}"] = {"wall_ms": round(ms, 3), "streams_over_the_frame": streams,
                                  "frac_of_hbm_peak": round(streams * gb / ms * 1e3 / HBM_PEAK_GBPS, 4)}
    if with_cpu:
        try:
            _report_cpu(out, xs, y, N, P)
        except Exception as e:
            out["report_c2_se"]["cpu"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            out["config1"] = _config1(torch, pds, ctx, dev)
        except Exception as e:
            out["config1"] = {"error": f"{type(e).__name__}: {e}"}
    n, p, w = 100_000_000, 8, 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    rx = [torch.rand(n, dtype=torch.float64, device=dev, generator=gen) for _ in range(p)]
    ry = sum(rx[j] * (0.1 * (j + 1)) for j in range(p)) + 1e-3 * torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    alg = n * ((p + 1) * 8 + (p + 1) * 8 + 8) / 1e9  # 152 B per row at p' = 8
    for name, fn in (("rolling_c4", lambda: pds.rolling_lin_reg(*rx, target=ry, window_size=w, ctx=ctx)),
                     ("expanding_1e8x8", lambda: pds.recursive_lin_reg(*rx, target=ry, start_with=w, ctx=ctx))):
        fn()
        ctx.get_timing(reset=True)
        ctx.set_timing(True)
        for _ in range(3):
            fn()
        ctx.set_timing(False)
        t = ctx.get_timing(reset=True)["rolling"]
        ms = t[0] / 3  # (the expanding fit is several launches per call)
        out[name] = {"rows": n, "coefficients": p, "window": w, "kernel_ms": round(ms, 3), "rows_per_s": round(n / ms * 1e3, 1),
                     "algorithmic_GBps": round(alg / ms * 1e3, 1), "frac_of_hbm_peak": round(alg / ms * 1e3 / HBM_PEAK_GBPS, 4)}
    ns = 2_000_000
    rx_s = [x[:ns].cpu().numpy() for x in rx]
    ry_s = ry[:ns].cpu().numpy()
    del rx, ry
    torch.cuda.empty_cache()
    try:
        out["elastic_net_c5"] = _c5(torch, pds, ctx, dev, with_cpu=with_cpu)
    except Exception as e:
        out["elastic_net_c5"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        import numpy as _np

        out["mid_width"] = _mid_width(torch, _np, pds, ctx, dev, wall)
    except Exception as e:
        out["mid_width"] = {"error": f"{type(e).__name__}: {e}"}
    try:  # the headline frame with one column (and sixteen columns) more: the step VERDICT r3 item 2 measures
        import numpy as _np

        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        more = [torch.randn(N, dtype=torch.float64, device=dev, generator=gen) for _ in range(16)]
        off = torch.arange(0, N + 1, 100, dtype=torch.int64, device=dev)
        cols = list(xs[:16]) + more
        steps = {}
        for q in (16, 17, 24, 32):
            steps[str(q)] = round(wall(lambda: pds.lin_reg_by(*cols[:q], target=y, group_offsets=off, add_bias=False, ctx=ctx)), 3)
        out["grouped_width_step"] = {"workload": f"{N // 100} groups x 100 rows, 16 / 17 / 24 / 32 f64 features (wall ms of lin_reg_by)", "ms": steps,
                                     "ratio_17_over_16": round(steps["17"] / steps["16"], 3), "ratio_32_over_16": round(steps["32"] / steps["16"], 3),
                                     "frac_of_hbm_peak": {k: round(N * (int(k) + 1) * 8 / (v * 1e-3) / 8e12, 3) for k, v in steps.items()}}
        # the widths in between, and f32 frames of the same shape (the reference's f32 twins, pl_lr_f32: linear_regression_f32.rs)
        more_steps = {}
        for q in (20, 28, 30):
            more_steps[str(q)] = round(wall(lambda: pds.lin_reg_by(*cols[:q], target=y, group_offsets=off, add_bias=False, ctx=ctx)), 3)
        out["grouped_width_step"]["ms_more"] = more_steps
        try:
            f32_cols = [c.float() for c in cols[:30]]
            y32 = y.float()
            pds.config.LIN_REG_EXPR_F64 = False
            out["grouped_width_step"]["ms_f32"] = {
                str(q): round(wall(lambda: pds.lin_reg_by(*f32_cols[:q], target=y32, group_offsets=off, add_bias=False, ctx=ctx)), 3) for q in (17, 24, 30)}
            del f32_cols, y32
        finally:
            pds.config.LIN_REG_EXPR_F64 = True
        del more, cols
        torch.cuda.empty_cache()
    except Exception as e:
        out["grouped_width_step"] = {"error": f"{type(e).__name__}: {e}"}
    if with_cpu:  # the reference's rolling driver is one sequential Woodbury chain: single thread, bounded sample of the same frame
        import numpy as np

        from oracle import oracle as orc

        Xh = np.stack(rx_s, axis=1)
        yh = ry_s
        t0 = time.perf_counter()
        orc.rolling_lr(Xh, yh, w)
        tc = time.perf_counter() - t0
        out["rolling_c4"]["cpu_rows_per_s"] = round(ns / tc, 1)
        out["rolling_c4"]["cpu_sample"] = f"{ns} rows of the same frame, the reference's sequential Woodbury chain (oracle port), 1 thread, {tc:.2f} s"
    return out


def _report_cpu(out, xs, y, N, P, ns=2_000_000):
    """pl_lin_reg_report on the host beside C2's report (linear_regression.rs:822-980): the oracle's restatement -- X'X, explicit inverse,
    beta = (inv X') y, residual pass (+ the leverage / meat sums of HC3) -- on a row PREFIX of the same frame with the ones column, one
    thread (the port is sequential; the reference's products run on faer's rayon pool), scaled by rows to the frame."""
    import numpy as np

    from oracle import oracle as orc

    ns = min(ns, N)
    X = np.empty((ns, P + 1), dtype=np.float64, order="F")
    for j in range(P):
        X[:, j] = xs[j][:ns].cpu().numpy()
    X[:, P] = 1.0
    yh = y[:ns].cpu().numpy()
    for se in ("se", "hc3"):
        t0 = time.perf_counter()
        rep = orc.lin_reg_report(X, yh, std_err=se)
        t = time.perf_counter() - t0
        gpu_s = out[f"report_c2_{se}"]["wall_ms"] / 1e3
        out[f"report_c2_{se}"]["cpu"] = {
            "kind": "port", "unit": "s/fit", "cores": 1, "value": round(t * N / ns, 2), "sample_s": round(t, 3),
            "sample": f"first {ns} rows of the same frame x {P} f64 features + ones column, std_err = {se}, 1 thread = {t:.2f} s; "
                      f"every pass is linear in rows: scaled by {N / ns:.0f} to the frame",
            "gpu_fit_s": round(gpu_s, 5), "gpu_over_cpu": round(t * N / ns / gpu_s, 1), "r2_of_sample": float(rep["r2"])}


def _config1(torch, pds, ctx, dev, calls=200):
    """configs[0] (benchmarks/test_linear_regression.py:9-31): pds.lin_reg(x1..x4, target=y, add_bias=False) on a 100 000-row f64 frame,
    seed 208 -- the reference's own CPU-runnable case.  Here: host Arrow buffers -> `_polars_plugin_pl_lr` -> Arrow list, through the
    plugin-ABI harness (PCIe both ways inside every call), the same columns resident in HBM through the C ABI, and the oracle's `pl_lr`
    (gated col-piv QR on X'X) on the host, 1 thread and all threads."""
    import numpy as np
    import pyarrow as pa

    sys.path.insert(0, str(ROOT / "tests"))
    import plugin_harness as ph
    from oracle import oracle as orc
    from polars_ds_extension_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(208)
    n = 100_000
    X = rng.random((n, 4))
    yh = X @ np.array([0.5, 0.25, -0.15, 0.2]) + 1e-4 * rng.random(n)
    host = [("y", pa.array(yh))] + [(f"x{j + 1}", pa.array(np.ascontiguousarray(X[:, j]))) for j in range(4)]
    kw = {"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
          "weighted": False, "positive": False, "singular_x_tol": 1e-12}

    def per_call(fn, k):
        for _ in range(5):
            r = fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            r = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k * 1e6, r

    us_plugin, res = per_call(lambda: ph.call_plugin(lib, "pl_lr", host, kw)[1], calls)
    b_gpu = np.asarray(res[0].as_py(), dtype=np.float64)
    dcols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).to(dev) for j in range(4)]
    dy = torch.from_numpy(yh).to(dev)
    us_dev, _ = per_call(lambda: pds.lin_reg(*dcols, target=dy, ctx=ctx), calls)
    Xf = np.asfortranarray(X)
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nt = min(nthreads, orc.max_threads()) if orc.max_threads() > 0 else nthreads
    cpu = {}
    for t in sorted({1, min(nt, 8)}):
        for _ in range(3):
            b_cpu = orc.solve_lr_gated(Xf, yh, 0.0, False, "qr", 1e-12, nthreads=t)
        t0 = time.perf_counter()
        for _ in range(50):
            b_cpu = orc.solve_lr_gated(Xf, yh, 0.0, False, "qr", 1e-12, nthreads=t)
        cpu[t] = (time.perf_counter() - t0) / 50 * 1e6
    best = min(cpu, key=cpu.get)
    return {"workload": "pds.lin_reg(x1..x4, target=y, add_bias=False), 100 000 rows f64, seed 208 (the reference's CPU-runnable benchmark case)",
            "plugin_abi_host_arrow_us_per_call": round(us_plugin, 1), "c_abi_hbm_resident_us_per_call": round(us_dev, 1),
            "bytes_per_call": n * 5 * 8, "hbm_resident_frac_of_hbm_peak": round(n * 5 * 8 / (us_dev * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "4 MB per call: launch + synchronisation latency, not bandwidth, on both GPU routes",
            "max_rel_err_vs_cpu": float(np.max(np.abs(b_gpu - b_cpu) / np.abs(b_cpu))),
            "cpu": {"kind": "port", "unit": "us/call", "cores": int(best), "value": round(cpu[best], 1),
                    "us_per_call_by_threads": {str(k): round(v, 1) for k, v in cpu.items()},
                    "sample": "the whole workload (100 000 rows x 4), 50 calls per thread count after 3 warm-up calls: X'X + X'y + rank gate + col-piv QR (pl_lr default path)"}}


def _c5(torch, pds, ctx, dev, with_cpu=True):
    """configs[4]: elastic net (l1 = l2 = 0.01, tol 1e-5) on 1e7 rows x 512 f32 features, AR(0.5) columns, 32 true coefficients --
    one Gram build (MFMA bound) + coordinate-descent sweeps on the 514 x 514 moment matrix.  Both f32 Gram arithmetics."""
    n, p = 10_000_000, 512
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    xs, prev = [], None
    for j in range(p):
        e = torch.randn(n, dtype=torch.float32, device=dev, generator=gen)
        prev = e if prev is None else 0.5 * prev + (0.75 ** 0.5) * e
        xs.append(prev)
    idx = torch.randperm(p, generator=torch.Generator().manual_seed(4))[:32]
    y = torch.zeros(n, dtype=torch.float32, device=dev)
    cgen = torch.Generator().manual_seed(5)
    for j in idx.tolist():
        y.add_(xs[j], alpha=float(torch.randn(1, generator=cgen).item()))
    y.add_(torch.randn(n, dtype=torch.float32, device=dev, generator=gen), alpha=0.5)
    out = {"rows": n, "features": p, "dtype": "f32"}
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        for name, native in (("bf16x3_split_default", "0"), ("f32_mfma", "1")):
            ctx.set_option("wide_f32_native", int(native))
            fit = lambda: pds.lin_reg(*xs, target=y, l1_reg=0.01, l2_reg=0.01, tol=1e-5, ctx=ctx)
            fit()
            ctx.get_timing(reset=True)
            ctx.set_timing(True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                b = fit()
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / 3 * 1e3
            ctx.set_timing(False)
            t = ctx.get_timing(reset=True)
            gram = t["moments"][0] / max(t["moments"][1], 1)
            useful = n * (p + 2) * (p + 3)  # flops of the upper triangle incl. the diagonal, 2 per multiply-add
            out[name] = {"wall_ms": round(wall, 2), "gram_ms": round(gram, 3), "cd_ms": round(t["iterative"][0] / max(t["iterative"][1], 1), 3),
                         "gram_useful_TFLOPs": round(useful / gram / 1e9, 1),
                         "frac_of_f32_mfma_peak_157TF": round(useful / gram / 1e9 / 157.3, 3), "nonzero": int((abs(b) > 1e-6).sum())}
            if native == "0":
                # the split path does NOT execute on the f32 pipe: every product is six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, hl,
                # lh, mm), so the pipe it runs on sees 6x the useful flops; against the dense bf16 peak (2.5 PFLOP/s) and against
                # the decomposition's own matrix-core floor (12.3 ms at the clock the kernel runs at)
                out[name]["executed_bf16_TFLOPs"] = round(6 * useful / gram / 1e9, 1)
                out[name]["frac_of_bf16_mfma_peak_2500TF"] = round(6 * useful / gram / 1e9 / 2500.0, 3)
                out[name]["frac_of_bf16_pipe_floor_12.3ms"] = round(12.3 / gram, 3)
                out[name]["pipe"] = "bf16 matrix cores (three exact bf16 planes per f32 value); the f32-peak fraction above is a useful-flop rate, not a utilisation of that pipe"
    finally:
        ctx.set_option("wide_f32_native", 0)
        pds.config.LIN_REG_EXPR_F64 = True
    if with_cpu:
        try:
            out["cpu"] = _c5_cpu(xs, y, n, p, out.get("bf16x3_split_default", {}).get("wall_ms"))
        except Exception as e:
            out["cpu"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _c5_cpu(xs, y, n, p, gpu_wall_ms, ns=200_000):
    """The reference's elastic net on the host beside C5 (benchmarks/test_linear_regression.py:119-179 times the same fit through sklearn
    and pds): the oracle's f32 `faer_coordinate_descent` (lr_solvers.rs:426-538) in its two parts -- X'X + X'y of a bounded ROW SAMPLE of
    the same frame on all host threads (the reference: faer matmul, Par::rayon(0)), scaled by rows to the frame; the covariance-update
    sweeps on the 512 x 512 block, one thread (sequential Gauss-Seidel, as the reference) -- they do not depend on the row count."""
    import numpy as np

    from oracle import oracle as orc

    orc.build()
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nt = min(nthreads, orc.max_threads()) if orc.max_threads() > 0 else nthreads
    X = np.empty((ns, p), dtype=np.float32, order="F")
    for j in range(p):
        X[:, j] = xs[j][:ns].cpu().numpy()
    yh = y[:ns].cpu().numpy()
    orc.gram(X[:4096], nthreads=nt)  # (warm the thread pool)
    t0 = time.perf_counter()
    G = orc.gram(X, nthreads=nt)
    c = orc.xty(X, yh, nthreads=nt)
    t_gram = time.perf_counter() - t0
    t0 = time.perf_counter()
    beta, sweeps, conv = orc.cd_from_gram(G, c, np.zeros(p, dtype=np.float32), 0.0, float(ns), 0.01, 0.01, False, 1e-5, 2000)
    t_cd = time.perf_counter() - t0
    gram_scaled = t_gram * (n / ns)
    res = {"kind": "port", "unit": "s/fit", "cores": nt, "nproc": os.cpu_count(),
           "sample": f"first {ns} rows of the same frame x {p} f32 features: X'X + X'y on {nt} threads = {t_gram:.2f} s (scaled by rows to {n:.0e}: "
                     f"{gram_scaled:.1f} s); {sweeps} covariance-update sweeps on the {p} x {p} block, 1 thread = {t_cd:.3f} s "
                     f"(f32 twin: tol 1e-5, at most 2000 sweeps, linear_regression_f32.rs:362)",
           "gram_sample_s": round(t_gram, 3), "gram_s_scaled_to_frame": round(gram_scaled, 2),
           "gram_GFLOPs_upper_triangle": round(ns * p * (p + 1) / t_gram / 1e9, 1),
           "sweeps": int(sweeps), "converged": bool(conv), "us_per_sweep": round(t_cd / max(sweeps, 1) * 1e6, 1), "cd_s": round(t_cd, 4),
           "value": round(gram_scaled + t_cd, 2), "nonzero": int((np.abs(beta) > 1e-6).sum())}
    if gpu_wall_ms:
        res["gpu_fit_s"] = round(gpu_wall_ms / 1e3, 4)
        res["gpu_over_cpu"] = round((gram_scaled + t_cd) / (gpu_wall_ms / 1e3), 1)
    return res


def _c3_spec(torch, pds, ctx, dev):
    sys.path.insert(0, str(ROOT / "tools"))
    import synth

    G, p = 1_000_000, 8
    fr = synth.c3_frame(G, p, seed=2, device=dev)
    xs, y, off = fr["xs"], fr["y"], fr["offsets"]

    def bench(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps, r

    t_sorted, (co, nu) = bench(lambda: pds.lin_reg_by(*xs, target=y, group_offsets=off, ctx=ctx))
    nulls = int(nu.sum().item())
    t_keyed, _ = bench(lambda: pds.lin_reg_by_key(*xs, target=y, key=fr["keys"], ctx=ctx, max_groups=G))
    perm = torch.randperm(fr["n_rows"], device=dev, generator=torch.Generator(device=dev).manual_seed(22))
    ks = fr["keys"][perm]
    xs_s = [x[perm] for x in xs]
    y_s = y[perm]
    del perm
    t_shuf, (k2, co2, nu2) = bench(lambda: pds.lin_reg_by_key(*xs_s, target=y_s, key=ks, ctx=ctx, max_groups=G), reps=3)
    alg = fr["n_rows"] * (p + 1) * 8 + G * (p * 8 + 17)
    return {"workload": f"{G} groups, Poisson(100) rows in [16, 256] ({fr['n_rows']} rows), {p} f64 feats, 0.1 % collinear groups",
            "gated_groups": nulls, "collinear_groups": int(fr["collinear"].sum().item()),
            "sorted_offsets_ms": round(t_sorted * 1e3, 3), "sorted_keys_ms": round(t_keyed * 1e3, 3), "shuffled_keys_ms": round(t_shuf * 1e3, 3),
            "regressions_per_s_sorted": round(G / t_sorted, 1), "regressions_per_s_shuffled": round(G / t_shuf, 1),
            "sorted_frac_of_hbm_peak": round(alg / t_sorted / 1e9 / HBM_PEAK_GBPS, 4),
            "shuffled_equals_sorted_nulls": bool((nu2 == nu).all().item())}


def _scatter_leg(torch, dist, par, dev, rank, world, G_total, R, P):
    """Rank 0 holds the whole frame; time scatter_frame_by_groups (grouped point-to-point, one launch)."""
    if rank == 0:
        xs_f, y_f = _gen_frame(torch, dev, 99, G_total, R, P)
        off_f = torch.arange(0, G_total * R + 1, R, dtype=torch.int64).numpy()
        args = (xs_f, y_f, off_f)
    else:
        args = (None, None, None)
    ts = []
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    for _ in range(2):
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        got = par.scatter_frame_by_groups(*args, root=0, device=dev)
        sync()
        dist.barrier()
        ts.append(time.perf_counter() - t0)
        del got
    t = torch.tensor([ts[-1]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sent = G_total * R * (P + 1) * 8 * (world - 1) / max(world, 1)
    return {"ms": round(float(t.item()) * 1e3, 2), "GB_sent_by_root": round(sent / 1e9, 2),
            "GBps_per_link": round(sent / max(world - 1, 1) / float(t.item()) / 1e9, 1) if world > 1 else None,
            "note": "frame resident on rank 0 only -> every rank holds its group range; not part of `value`"}


if __name__ == "__main__":
    sys.exit(main())
