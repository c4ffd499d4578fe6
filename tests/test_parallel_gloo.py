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
        out = {"rank": rank, "err_rows": err_rows, "range": (g_lo, g_hi)}
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
