"""
world_size-2 gloo tests (CPU) of the multi-GPU orchestration: row-sharded single regression (one all-reduce
of the moment block) and group-sharded regressions (no data-path collective, results gathered to rank 0).
The compute steps are injected with the CPU oracle, so only the sharding / collective logic is under test.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc
    from polars_ds_extension_amd import parallel as par

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)  # same frame on every rank
        n, p = 10_007, 5
        X = rng.random((n, p))
        y = X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)
        # ---- row-sharded single OLS
        lo, hi = par.shard_bounds(n, world, rank)

        def moments_fn(xs, yy, w):
            Z = np.c_[np.stack(xs, axis=1), np.ones(len(yy)), yy]
            return torch.from_numpy(np.ascontiguousarray(orc.gram(Z)))

        def solve_fn(M, add_bias=False, **kw):
            M = M.numpy()
            pp = p + int(add_bias)
            return orc.gated_solve_gram(np.ascontiguousarray(M[:pp, :pp]), M[:pp, p + 1].copy(), tol=1e-12)

        b = par.lin_reg_row_sharded([X[lo:hi, j] for j in range(p)], y[lo:hi], moments_fn=moments_fn, solve_fn=solve_fn,
                                    add_bias=True)
        ref = orc.pl_lr(X, y, add_bias=True)
        err_rows = float(np.linalg.norm(b - ref) / np.linalg.norm(ref))
        # ---- group-sharded
        sizes = rng.integers(8, 60, size=300)
        off = np.concatenate([[0], np.cumsum(sizes)])[: 1 + np.searchsorted(np.cumsum(sizes), n, side="right")]
        ng = len(off) - 1

        def grouped_fn(xs, yy, loc_off, add_bias=False, **kw):
            co, nu = orc.grouped_lr([np.ascontiguousarray(yy)] + [np.ascontiguousarray(x) for x in xs], loc_off, add_bias=add_bias)
            return torch.from_numpy(co), torch.from_numpy(nu.astype(np.uint8))

        g_lo, g_hi, co, nu = par.lin_reg_by_group_sharded([X[:, j] for j in range(p)], y, off, grouped_fn=grouped_fn,
                                                          gather_to=0, add_bias=False)
        # ---- the frame lives on rank 0 only: scatter leg (grouped point-to-point), then compute + gather in 3 pieces per rank,
        #      gathered on rank 1 this time (the root of the scatter and the root of the gather need not coincide)
        n_used = int(off[-1])
        if rank == 0:
            xs_t = [torch.from_numpy(np.ascontiguousarray(X[:n_used, j])) for j in range(p)]
            sc = par.scatter_frame_by_groups(xs_t, torch.from_numpy(y[:n_used].copy()), off, root=0)
        else:
            sc = par.scatter_frame_by_groups(None, None, None, root=0, device=torch.device("cpu"))
        xs_l, y_l, off_l, parts = sc

        def grouped_fn_t(xs, yy, loc_off, add_bias=False, **kw):
            return grouped_fn([np.asarray(x) for x in xs], np.asarray(yy), np.asarray(loc_off), add_bias=add_bias)

        res = par.lin_reg_by_group_local_shard(xs_l, y_l, off_l, parts, rank=rank, gather_to=1, chunks=3, grouped_fn=grouped_fn_t,
                                               add_bias=True)
        # ---- the prepared plan of the same leg: persistent result buffers, the gathering rank's shard fitted in place, one grouped
        #      send per piece; two steps give the same (views of the same) results; then the orchestration alone is timed with a
        #      compute step that does nothing (what a step costs beside the kernels: Python + the point-to-point launches)
        plan = par.GroupedShardPlan(xs_l, y_l, off_l, parts, rank=rank, gather_to=1, chunks=3, grouped_fn=grouped_fn_t, add_bias=True)
        st1 = plan.step()
        first = [t.clone() for t in st1]
        st2 = plan.step()
        plan_ok = bool(all(a.data_ptr() == b.data_ptr() for a, b in zip(st1, st2)) and
                       all(torch.equal(a, b) for a, b in zip(first, st2)) and
                       all(torch.equal(a, b) for a, b in zip(first, res)))
        if rank == 1:  # in place: the gathering rank's local results ARE rows of the assembled result
            lo_g = parts[1][0]
            plan_ok = plan_ok and st2[0].data_ptr() == st2[2][lo_g:].data_ptr() and st2[1].data_ptr() == st2[3][lo_g:].data_ptr()
        pp_n = p + 1

        def noop_fn(xs, yy, loc_off, add_bias=False, **kw):
            k = len(loc_off) - 1
            return torch.zeros((k, pp_n), dtype=torch.float64), torch.zeros(k, dtype=torch.uint8)

        import time as _time

        lean = par.GroupedShardPlan(xs_l, y_l, off_l, parts, rank=rank, gather_to=1, chunks=1, grouped_fn=noop_fn, add_bias=True)
        for _ in range(5):
            lean.step()
        dist.barrier()
        ts = []
        for _ in range(30):
            t0 = _time.perf_counter()
            lean.step()
            ts.append(_time.perf_counter() - t0)
        plan_step_ms = 1e3 * sorted(ts)[len(ts) // 2]
        auto_c = par.GroupedShardPlan(xs_l, y_l, off_l, parts, rank=rank, gather_to=1, grouped_fn=noop_fn, add_bias=True).chunks
        sc_rows_ok = bool(int(off_l[-1]) == len(y_l) and int(off_l[0]) == 0 and
                          np.array_equal(np.asarray(y_l), y[int(off[parts[rank][0]]): int(off[parts[rank][1]])]))
        # ---- lin_reg_report, row-sharded: two all-reduces (moment block; [sum e^2 | sum w e^2 | meat])
        def rep_moments_fn(xs, yy, w):
            Z = np.c_[np.stack(xs, axis=1), np.ones(len(yy)), yy]
            return torch.from_numpy(np.ascontiguousarray(Z.T @ (Z if w is None else Z * np.asarray(w)[:, None])))

        def rep_fit_fn(M, add_bias=False):
            pp_ = p + int(add_bias)
            G = np.ascontiguousarray(M[:pp_, :pp_])
            inv = orc.qr_inverse(G)
            return inv @ M[:pp_, p + 1], inv

        def rep_partials_fn(xs, yy, w, beta, inv, add_bias=False, std_err="se"):
            Xl = np.stack(xs, axis=1)
            if add_bias:
                Xl = np.c_[Xl, np.ones(len(yy))]
            e = yy - Xl @ beta
            h = np.einsum("ij,jk,ik->i", Xl, inv, Xl)
            s2 = {"se": np.zeros_like(e), "hc0": e * e, "hc1": e * e, "hc2": e * e / (1 - h), "hc3": e * e / (1 - h) ** 2}[std_err]
            q_ = p + 2
            meat = np.zeros((q_, q_))
            idx = list(range(p)) + ([p] if add_bias else [])
            meat[np.ix_(idx, idx)] = Xl.T @ (Xl * s2[:, None])
            ww = np.ones_like(e) if w is None else np.asarray(w)
            return np.r_[e @ e, (ww * e) @ e, meat.reshape(-1, order="F")]

        def rep_finish_fn(beta, inv, part, n_rows_total, y_var, add_bias=False, weighted=False, std_err="se"):
            pp_ = len(beta)
            q_ = p + 2
            dof = n_rows_total - pp_
            if std_err == "se":
                se = np.sqrt(part[0] / dof * np.diag(inv))
            else:
                meat = part[2:].reshape(q_, q_, order="F")
                idx = list(range(p)) + ([p] if add_bias else [])
                V = inv @ meat[np.ix_(idx, idx)] @ inv
                se = np.sqrt(np.diag(V) * (n_rows_total / dof if std_err == "hc1" else 1.0))
            return {"beta": beta, "se": se, "r2": 1 - part[0] / (y_var * n_rows_total)}

        rep_err = {}
        Xb_full = np.c_[X, np.ones(n)]
        for se_kind in ("se", "hc1", "hc3"):
            rep = par.lin_reg_report_row_sharded([X[lo:hi, j] for j in range(p)], y[lo:hi], add_bias=True, std_err=se_kind,
                                                 moments_fn=rep_moments_fn, fit_fn=rep_fit_fn, partials_fn=rep_partials_fn,
                                                 finish_fn=rep_finish_fn)
            ro = orc.lin_reg_report(Xb_full, y, std_err=se_kind)
            rep_err[se_kind] = (float(np.max(np.abs(rep["se"] - ro["std_err"]) / ro["std_err"])),
                                float(np.linalg.norm(rep["beta"] - ro["beta"]) / np.linalg.norm(ro["beta"])), abs(rep["r2"] - ro["r2"]))
        # ---- rolling, row-sharded with a (window - 1)-row halo; compute step = the sequential oracle
        win = 37

        def rolling_fn(xs, yy, window_size, add_bias=False, **kw):
            Xl = np.stack(xs, axis=1)
            if add_bias:
                Xl = np.c_[Xl, np.ones(len(yy))]
            co = orc.rolling_lr(Xl, yy, window_size)
            full = np.full((len(yy), Xl.shape[1]), np.nan)
            full[window_size - 1 :] = co
            va = np.zeros(len(yy), dtype=np.uint8)
            va[window_size - 1 :] = 1
            return full, np.einsum("ij,ij->i", Xl, np.nan_to_num(full)), va

        nr = 2003
        r_lo, r_hi, rco, rpr, rva = par.rolling_lin_reg_row_sharded([X[:nr, j] for j in range(p)], y[:nr], win,
                                                                      rolling_fn=rolling_fn, add_bias=True)
        ref_roll = orc.rolling_lr(np.c_[X[:nr], np.ones(nr)], y[:nr], win)  # rows win-1 .. nr-1
        first = max(r_lo, win - 1)
        err_roll = float(np.max(np.abs(rco[first - r_lo :] - ref_roll[first - (win - 1) : r_hi - (win - 1)])))
        roll_valid_ok = bool(np.all(rva[first - r_lo :] == 1) and np.all(rva[: first - r_lo] == 0))

        # ---- expanding ("recursive"), row-sharded: all-gather of moment blocks, exclusive prefix seeds each rank
        n0 = 11

        def rec_moments_fn(xs, yy, w):
            Z = np.c_[np.stack(xs, axis=1), np.ones(len(yy)), yy]
            return torch.from_numpy(np.ascontiguousarray(Z.T @ Z))

        def recursive_fn(xs, yy, start_with, seed_moments=None, add_bias=False, **kw):
            Xl = np.stack(xs, axis=1)
            k = Xl.shape[1]
            if add_bias:
                Xl = np.c_[Xl, np.ones(len(yy))]
            pp_ = Xl.shape[1]
            G = np.zeros((pp_, pp_)); c = np.zeros(pp_); seen = 0.0
            if seed_moments is not None:
                M = seed_moments.numpy()
                idx = list(range(k)) + ([k] if add_bias else [])
                G = M[np.ix_(idx, idx)].copy(); c = M[idx, k + 1].copy(); seen = M[k, k]
            out = np.full((len(yy), pp_), np.nan); va = np.zeros(len(yy), dtype=np.uint8)
            for i in range(len(yy)):
                G += np.outer(Xl[i], Xl[i]); c += Xl[i] * yy[i]; seen += 1
                if seen >= start_with:
                    out[i] = np.linalg.solve(G, c); va[i] = 1
            return out, None, va

        e_lo, e_hi = par.shard_bounds(nr, world, rank)
        eco, _, eva = par.recursive_lin_reg_row_sharded([X[e_lo:e_hi, j] for j in range(p)], y[e_lo:e_hi], n0,
                                                        moments_fn=rec_moments_fn, recursive_fn=recursive_fn, add_bias=True)
        ref_rec = orc.recursive_lr(np.c_[X[:nr], np.ones(nr)], y[:nr], n0)  # rows n0-1 .. nr-1
        first = max(e_lo, n0 - 1)
        err_rec = float(np.max(np.abs(eco[first - e_lo :] - ref_rec[first - (n0 - 1) : e_hi - (n0 - 1)])))
        rec_valid_ok = bool(np.all(eva[first - e_lo :] == 1) and np.all(eva[: first - e_lo] == 0))
        out = {"rank": rank, "err_rows": err_rows, "range": (g_lo, g_hi), "err_roll": err_roll, "roll_valid_ok": roll_valid_ok,
               "err_rec": err_rec, "rec_valid_ok": rec_valid_ok, "sc_rows_ok": sc_rows_ok, "rep_err": rep_err,
               "sc_local_groups": int(res[0].shape[0]), "parts": parts, "plan_ok": plan_ok, "plan_step_ms": plan_step_ms,
               "auto_chunks_small": auto_c}
        if rank == 1:
            ref_co_b, ref_nu_b = orc.grouped_lr([y[:n_used]] + [X[:n_used, j] for j in range(p)], off, add_bias=True)
            out["sc_gathered"] = tuple(res[2].shape)
            out["sc_err"] = float(np.max(np.abs(res[2].numpy() - ref_co_b)))
            out["sc_null_equal"] = bool(np.array_equal(res[3].numpy().astype(bool), ref_nu_b))
        if rank == 0:
            ref_co, ref_nu = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off)
            out["groups"] = ng
            out["gathered"] = tuple(co.shape)
            out["err_groups"] = float(np.max(np.abs(co.numpy() - ref_co)))
            out["null_equal"] = bool(np.array_equal(nu.numpy().astype(bool), ref_nu))
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_shard_helpers():
    sys.path.insert(0, str(ROOT))
    from polars_ds_extension_amd import parallel as par

    for n, w in ((10, 3), (7, 8), (0, 2), (1_000_000, 8)):
        b = [par.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    off = np.concatenate([[0], np.cumsum(np.random.default_rng(1).integers(1, 500, size=1000))])
    parts = par.shard_groups_by_rows(off, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 1000 and all(parts[i][1] == parts[i + 1][0] for i in range(7))
    rows = [off[h] - off[l] for l, h in parts]
    assert max(rows) - min(rows) < 2 * 500  # balanced in rows up to one group
    # pieces per shard: the headline frame over 8 ranks pipelines in 4 pieces, a world of one (or tiny results) in one
    assert par.auto_chunks(8, 125_000 * (16 * 8 + 1), 12_500_000 * 17 * 8) == 4
    assert par.auto_chunks(1, 10**9, 10**10) == 1 and par.auto_chunks(8, 4096, 10**6) == 1
    assert par.chunk_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)] and par.chunk_bounds(0, 4) == [(0, 0)]


@pytest.mark.timeout(300)
def test_world2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    outs = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    outs.sort(key=lambda o: o["rank"])
    assert all(o["err_rows"] < 1e-11 for o in outs)  # all-reduced moments -> same solution on every rank
    assert outs[0]["range"][1] == outs[1]["range"][0] and outs[0]["range"][0] == 0 and outs[1]["range"][1] == outs[0]["groups"]
    assert outs[0]["gathered"][0] == outs[0]["groups"] and outs[0]["err_groups"] < 1e-12 and outs[0]["null_equal"]
    # halo-sharded rolling == the sequential chain over the whole frame (the chain of rank 1 starts at its halo,
    # so it differs from the reference's never re-anchored chain by that chain's own round-off only)
    assert all(o["err_roll"] < 1e-8 and o["roll_valid_ok"] for o in outs)
    assert all(o["err_rec"] < 1e-8 and o["rec_valid_ok"] for o in outs)
    # scatter from rank 0, three pieces per rank, gathered on rank 1: every rank got exactly its rows, rank 1 holds all groups
    assert all(o["sc_rows_ok"] for o in outs)
    parts = outs[0]["parts"]
    assert [o["sc_local_groups"] for o in outs] == [hi - lo for lo, hi in parts]
    assert outs[1]["sc_gathered"][0] == outs[0]["groups"] and outs[1]["sc_err"] < 1e-12 and outs[1]["sc_null_equal"]
    # the prepared plan: same results step after step out of the same buffers, in place on the gathering rank; its orchestration
    # (no compute) stays far below a millisecond even over gloo / TCP loopback (RCCL world 1 on the GPU: bench.py dist_step_overhead_us)
    assert all(o["plan_ok"] for o in outs)
    assert all(o["plan_step_ms"] < 5.0 for o in outs), [o["plan_step_ms"] for o in outs]
    assert all(o["auto_chunks_small"] == 1 for o in outs)  # a few KB of results: nothing to overlap
    # row-sharded report == the single-frame report (both all-reduces), on every rank
    for o in outs:
        for kind, (e_se, e_beta, e_r2) in o["rep_err"].items():
            assert e_se < 1e-10 and e_beta < 1e-11 and e_r2 < 1e-12, (kind, e_se, e_beta, e_r2)
