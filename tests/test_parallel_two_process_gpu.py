"""
The DIRECT gather of the group-sharded step (parallel.GroupedShardPlan(direct=True), include/pds_lstsq.h pds_device_alloc / pds_ipc_*):
a peer PROCESS's fused kernel stores its coefficients and flags straight into the gathering rank's result block.  A test box has one
GPU, so the two ranks are two processes on that one device (the IPC mapping, the raw output addresses, the completion all-reduce and
the assembled result are the real ones; what is not exercised is the xGMI hop) with gloo as the control plane (RCCL refuses two ranks
on one device).  Completion is the stream-ordered word per peer (pds_signal_post / pds_signal_wait), no collective in a step.  With two visible devices the second rank moves to cuda:1 and the stores cross the link.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, direct=True):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import torch
        import torch.distributed as dist

        import polars_ds_extension_amd as pds
        from polars_ds_extension_amd import parallel as par

        ndev = torch.cuda.device_count()
        dev_i = rank % ndev
        torch.cuda.set_device(dev_i)
        dev = torch.device("cuda", dev_i)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = pds.Context(dev_i)
        rng = np.random.default_rng(31)  # the same frame on both ranks
        p, G = 16, 6000
        sizes = rng.integers(20, 140, size=G)
        sizes[100] = 5  # fewer rows than coefficients: a null group
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(off[-1])
        X = rng.normal(size=(n, p))
        y = X @ rng.normal(size=p) + 0.1 * rng.normal(size=n)
        parts = par.shard_groups_by_rows(off, world)
        g_lo, g_hi = parts[rank]
        r_lo, r_hi = int(off[g_lo]), int(off[g_hi])
        xs_loc = [torch.from_numpy(np.ascontiguousarray(X[r_lo:r_hi, j])).to(dev) for j in range(p)]
        y_loc = torch.from_numpy(y[r_lo:r_hi].copy()).to(dev)
        loc_off = torch.from_numpy(off[g_lo:g_hi + 1] - r_lo).to(dev)
        if direct:
            plan = par.GroupedShardPlan(xs_loc, y_loc, loc_off, parts, rank=rank, gather_to=0, direct=True, ctx=ctx, add_bias=False)
        else:  # the default form: prepared fits in three pieces writing in place, a piece's results sent while the next is fitted
            plan = par.GroupedShardPlan(xs_loc, y_loc, loc_off, parts, rank=rank, gather_to=0, chunks=3, ctx=ctx, add_bias=False)
        res = {"rank": rank, "device": dev_i}
        for k in range(3):
            if rank == 0 and k > 0:
                plan.co_all.fill_(123.0)  # (a step must rewrite every row, the peers' included)
                torch.cuda.synchronize()
            dist.barrier()
            out = plan.step()
        torch.cuda.synchronize()
        # what this rank's shard gives through the plain call (same slices: same tile alignment -> the same bits; the pieces of the
        # point-to-point form start at other rows: equal to rounding there)
        co_ref, nu_ref = pds.lin_reg_by(*xs_loc, target=y_loc, group_offsets=loc_off, add_bias=False, ctx=ctx)
        res["local_ref"] = (co_ref.cpu().numpy(), nu_ref.cpu().numpy())
        res["direct"] = direct
        if rank == 0:
            res["assembled"] = (out[2].cpu().numpy(), out[3].cpu().numpy())
            res["views_in_place"] = bool(out[0].data_ptr() == out[2][g_lo:].data_ptr())
        else:
            res["peer_returns_none"] = out == (None, None) if direct else (out[0].shape[0] == g_hi - g_lo)
        res["parts"] = parts
        res["wait_timeouts"] = ctx.signal_wait_timeouts() if direct else 0
        dist.barrier()
        plan.close()  # (collective: the peer unmaps, then the owner frees)
        q.put(res)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent turns this into a failure (or a skip when the box cannot map IPC memory)
        import traceback

        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


@pytest.mark.timeout(300)
@pytest.mark.parametrize("direct", [True, False], ids=["direct-stores", "point-to-point"])
def test_gather_two_processes(direct):
    """direct: the peers' kernels store into rank 0's block.  point-to-point: the DEFAULT N-rank step with device-resident shards and the
    library's prepared fits (what `bench.py --gpus N` runs), its sends / receives carried by gloo here instead of RCCL."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, direct)) for r in range(2)]
    for pr in procs:
        pr.start()
    outs = []
    for _ in procs:
        outs.append(q.get(timeout=240))
        if "error" in outs[-1]:
            break
    for pr in procs:
        pr.join(timeout=30)
        if pr.is_alive():
            pr.kill()
    errs = [o for o in outs if "error" in o]
    if errs and "hipIpc" in errs[0]["error"] + errs[0]["trace"]:
        pytest.skip(f"this box does not map IPC memory between processes: {errs[0]['error']}")
    assert not errs, errs[0]["trace"]
    outs.sort(key=lambda o: o["rank"])
    parts = outs[0]["parts"]
    co, nu = outs[0]["assembled"]
    assert outs[0]["views_in_place"] and outs[1]["peer_returns_none"]
    assert outs[0]["wait_timeouts"] == 0  # every step's wait saw the peer's word
    for o in outs:
        lo, hi = parts[o["rank"]]
        if direct:
            assert np.array_equal(co[lo:hi], o["local_ref"][0], equal_nan=True), o["rank"]
        else:
            assert np.allclose(co[lo:hi], o["local_ref"][0], rtol=1e-12, atol=1e-14, equal_nan=True), o["rank"]
        assert np.array_equal(nu[lo:hi], o["local_ref"][1]), o["rank"]
    assert nu.sum() == 1 and nu[100] == 1 and not np.any(co[~nu.astype(bool)] == 123.0)
