"""
world_size-8 gloo rehearsal (CPU, oracle-injected compute) of the group-sharded step as the 8-GPU bench runs it: a GroupedShardPlan whose
piece count comes from the cost model and is agreed by the all-reduce (shards that differ by one group must not round it differently),
4 pieces, a gathering rank that is not rank 0, three consecutive steps out of the same buffers, bit-equal to the single-rank result;
and the scatter leg with empty shards (fewer groups than ranks).
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
WORLD = 8


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
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch
    import torch.distributed as dist

    torch.set_num_threads(1)
    from oracle import oracle as orc
    from polars_ds_extension_amd import parallel as par

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(8)  # the same frame on every rank
        p, G = 4, 2003                  # 2003 = 8 x 250 + 3: the first three shards hold one group more
        sizes = rng.integers(8, 40, size=G)
        sizes[17] = 3                   # fewer rows than coefficients with the intercept: a null group in the middle of a shard
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(off[-1])
        X = rng.normal(size=(n, p))
        y = X @ rng.normal(size=p) + 0.5 + 0.1 * rng.normal(size=n)
        parts = [par.shard_bounds(G, world, r) for r in range(world)]
        g_lo, g_hi = parts[rank]
        r_lo, r_hi = int(off[g_lo]), int(off[g_hi])
        xs_loc = [torch.from_numpy(np.ascontiguousarray(X[r_lo:r_hi, j])) for j in range(p)]
        y_loc = torch.from_numpy(y[r_lo:r_hi].copy())
        loc_off = torch.from_numpy(off[g_lo:g_hi + 1] - r_lo)

        def grouped_fn(xs, yy, o, add_bias=False, **kw):
            co, nu = orc.grouped_lr([np.ascontiguousarray(np.asarray(yy))] + [np.ascontiguousarray(np.asarray(x)) for x in xs],
                                    np.asarray(o), add_bias=add_bias)
            return torch.from_numpy(co), torch.from_numpy(nu.astype(np.uint8))

        # the cost model, with a launch cost chosen so that the LARGEST shard's estimate is exactly 4 pieces: the smaller shards
        # estimate <= 4 on their own, the agreement all-reduce (MAX) makes it 4 everywhere
        pp = p + 1
        t_x = max(hi - lo for r, (lo, hi) in enumerate(parts) if r != 3) * (pp * 8 + 1) / 45e9
        t_k = max(int(off[hi] - off[lo]) for lo, hi in parts) * (p + 1) * 8 / 5500e9
        model = {"launch_us": min(t_x, t_k) / 16 * 1e6}
        own = par.auto_chunks(world, max(hi - lo for r, (lo, hi) in enumerate(parts) if r != 3) * (pp * 8 + 1),
                              int(off[g_hi] - off[g_lo]) * (p + 1) * 8, **model)
        plan = par.GroupedShardPlan(xs_loc, y_loc, loc_off, parts, rank=rank, gather_to=3, grouped_fn=grouped_fn, chunk_model=model,
                                    add_bias=True)
        outs = [plan.step() for _ in range(3)]
        same_buffers = all(a.data_ptr() == b.data_ptr() for st in outs[1:] for a, b in zip(outs[0], st))
        res = {"rank": rank, "chunks": plan.chunks, "own_estimate": own, "same_buffers": bool(same_buffers),
               "local_groups": int(outs[-1][0].shape[0])}
        ref_co, ref_nu = orc.grouped_lr([y] + [np.ascontiguousarray(X[:, j]) for j in range(p)], off, add_bias=True)
        res["local_equal"] = bool(np.array_equal(outs[-1][0].numpy(), ref_co[g_lo:g_hi], equal_nan=True) and
                                  np.array_equal(outs[-1][1].numpy().astype(bool), ref_nu[g_lo:g_hi]))
        if rank == 3:
            co, nu = outs[-1][2].numpy(), outs[-1][3].numpy().astype(bool)
            res["gathered_equal"] = bool(np.array_equal(co, ref_co, equal_nan=True) and np.array_equal(nu, ref_nu))
            res["gathered_shape"] = tuple(co.shape)
            res["null_groups"] = int(nu.sum())
        # ---- scatter leg with EMPTY shards: 5 groups over 8 ranks, frame resident on rank 2
        Gs = 5
        off_s = off[: Gs + 1]
        ns = int(off_s[-1])
        if rank == 2:
            sc = par.scatter_frame_by_groups([torch.from_numpy(np.ascontiguousarray(X[:ns, j])) for j in range(p)],
                                             torch.from_numpy(y[:ns].copy()), off_s, root=2)
        else:
            sc = par.scatter_frame_by_groups(None, None, None, root=2, device=torch.device("cpu"))
        xs_s, y_s, off_l, parts_s = sc
        lo_s, hi_s = parts_s[rank]
        res["scatter_parts"] = parts_s
        res["scatter_ok"] = bool(len(y_s) == int(off_s[hi_s] - off_s[lo_s]) and
                                 np.array_equal(np.asarray(y_s), y[int(off_s[lo_s]): int(off_s[hi_s])]) and
                                 np.array_equal(np.asarray(off_l), off_s[lo_s:hi_s + 1] - off_s[lo_s]))
        # ... and the plan over those shards (empty ones included), gathered on rank 0
        plan_s = par.GroupedShardPlan(xs_s, y_s, off_l, parts_s, rank=rank, gather_to=0, chunks=2, grouped_fn=grouped_fn, add_bias=False)
        st = plan_s.step()
        if rank == 0:
            ref_s, nu_s = orc.grouped_lr([y[:ns]] + [np.ascontiguousarray(X[:ns, j]) for j in range(p)], off_s)
            res["scatter_gathered_equal"] = bool(np.array_equal(st[2].numpy(), ref_s) and np.array_equal(st[3].numpy().astype(bool), nu_s))
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world8_gloo_grouped_plan_and_scatter():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for pr in procs:
        pr.start()
    outs = [q.get(timeout=480) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    outs.sort(key=lambda o: o["rank"])
    assert [o["chunks"] for o in outs] == [4] * WORLD, [(o["own_estimate"], o["chunks"]) for o in outs]  # agreed, = the model's 4
    assert max(o["own_estimate"] for o in outs) == 4
    assert [o["local_groups"] for o in outs] == [251, 251, 251, 250, 250, 250, 250, 250]
    assert all(o["same_buffers"] and o["local_equal"] for o in outs)
    assert outs[3]["gathered_equal"] and outs[3]["gathered_shape"] == (2003, 5) and outs[3]["null_groups"] == 1
    parts_s = outs[0]["scatter_parts"]
    assert sum(1 for lo, hi in parts_s if hi == lo) >= 3  # empty shards took part
    assert all(o["scatter_ok"] for o in outs)
    assert outs[0]["scatter_gathered_equal"]
