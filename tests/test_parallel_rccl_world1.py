"""
The multi-GPU orchestration with its DEFAULT compute steps (the HIP library) and the RCCL backend, at world_size 1 on
the one GPU a test box has: init / all-reduce / all-gather / gather run through RCCL, the sharding arithmetic through
its rank-0-of-1 case.  (World-2 logic is covered on CPU with gloo in test_parallel_gloo.py.)
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def world1():
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as e:  # an environment without a usable RCCL is not a failure of this library
        pytest.skip(f"RCCL process group could not be created: {e}")
    yield dist
    dist.destroy_process_group()


def test_sharded_paths_on_rccl(world1):
    import torch

    import polars_ds_extension_amd as pds
    from polars_ds_extension_amd import parallel as par

    rng = np.random.default_rng(3)
    n, p = 60_000, 6
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)
    xs = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]
    yt = torch.from_numpy(y).cuda()
    # row-sharded single regression: moments -> all-reduce -> replicated solve
    b = par.lin_reg_row_sharded(xs, yt, add_bias=True)
    b0 = pds.lin_reg(*xs, target=yt, add_bias=True)
    assert np.linalg.norm(np.asarray(b) - b0) / np.linalg.norm(b0) < 1e-12
    # group-sharded, gathered on rank 0
    sizes = rng.integers(10, 200, size=500)
    off = np.concatenate([[0], np.cumsum(sizes)])
    off = off[off <= n]
    g_lo, g_hi, co_all, nu_all = par.lin_reg_by_group_sharded(xs, yt, off, gather_to=0, add_bias=False)
    ref_co, ref_nu = pds.lin_reg_by(*xs, target=yt, group_offsets=off)
    assert (g_lo, g_hi) == (0, len(off) - 1)
    assert torch.allclose(co_all.to(ref_co.device), ref_co, rtol=0, atol=0, equal_nan=True)
    assert torch.equal(nu_all.to(ref_nu.device), ref_nu)
    # the same through the scatter leg + compute / gather in pieces (rank 0 of 1: the grouped send / recv lists are empty, the
    # partition, piece bounds and result assembly are the real ones)
    xs_l, y_l, off_l, parts = par.scatter_frame_by_groups(xs, yt[: int(off[-1])], off, root=0)
    co_l, nu_l, co_g, nu_g = par.lin_reg_by_group_local_shard(xs_l, y_l, off_l, parts, rank=0, gather_to=0, chunks=4, add_bias=False)
    # (pieces start at other rows than the frame, so a group's rows meet the kernel's 128-row tiles differently: same sums in
    #  another order -- equal to rounding, not bitwise)
    assert parts == [(0, len(off) - 1)] and torch.allclose(co_g, ref_co, rtol=1e-12, atol=1e-14, equal_nan=True) and torch.equal(nu_g, ref_nu)
    # the prepared plan (what bench.py's N-rank step runs): the library's fit as prepared calls writing IN PLACE into the rows of the
    # persistent assembled result; step after step the same buffers, the same values
    plan = par.GroupedShardPlan(xs_l, y_l, off_l, parts, rank=0, gather_to=0, chunks=3, add_bias=False)
    s1 = plan.step()
    keep = [t.clone() for t in s1]
    s1[2].zero_()
    s2 = plan.step()
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(s1, s2)) and s2[0].data_ptr() == s2[2].data_ptr()
    assert all(torch.equal(a, b) for a, b in zip(keep, s2))
    assert torch.allclose(s2[2], ref_co, rtol=1e-12, atol=1e-14, equal_nan=True) and torch.equal(s2[3], ref_nu)
    one = par.GroupedShardPlan(xs_l, y_l, off_l, parts, rank=0, gather_to=0, add_bias=False)  # (auto: one piece at world 1)
    assert one.chunks == 1 and torch.allclose(one.step()[2], ref_co, rtol=1e-12, atol=1e-14, equal_nan=True)
    fit = pds.GroupedFit(*xs, target=yt, group_offsets=off)
    co_f, nu_f = fit.run()
    assert torch.equal(co_f, ref_co) and torch.equal(nu_f, ref_nu) and fit.run()[0].data_ptr() == co_f.data_ptr()
    with pytest.raises(ValueError):
        pds.GroupedFit(*xs, target=yt, group_offsets=off, out=torch.empty((3, p), dtype=torch.float64, device="cuda"))
    # row-sharded report: moments -> all-reduce -> fit -> residual pass -> all-reduce -> epilogue == the single-frame report
    for kind in ("se", "hc0", "hc1", "hc2", "hc3"):
        rs = par.lin_reg_report_row_sharded(xs, yt, add_bias=True, std_err=kind)
        r0 = pds.lin_reg_report(*xs, target=yt, add_bias=True, std_err=kind)
        for k in r0:
            if k != "features":
                assert np.allclose(rs[k], r0[k], rtol=1e-13, atol=0), (kind, k)
    w = torch.from_numpy(rng.random(n) + 0.5).cuda()
    yv = float(np.var(y, ddof=1))
    rs = par.lin_reg_report_row_sharded(xs, yt, add_bias=True, weights_local=w, y_var=yv)
    r0 = pds.lin_reg_report(*xs, target=yt, add_bias=True, weights=w, y_var=yv)
    assert all(np.allclose(rs[k], r0[k], rtol=1e-13, atol=0) for k in r0 if k != "features")
    # rolling with halo / expanding with an all-gathered prefix
    lo, hi, rc, rp, rv = par.rolling_lin_reg_row_sharded(xs, yt, 64, add_bias=True)
    c0, p0, v0 = pds.rolling_lin_reg(*xs, target=yt, window_size=64, add_bias=True)
    assert (lo, hi) == (0, n) and torch.equal(rv, v0) and torch.allclose(rc[63:], c0[63:], rtol=0, atol=0)
    ec, ep, ev = par.recursive_lin_reg_row_sharded(xs, yt, 12, add_bias=True)
    c1, p1, v1 = pds.recursive_lin_reg(*xs, target=yt, start_with=12, add_bias=True)
    assert torch.equal(ev, v1) and torch.allclose(ec[11:], c1[11:], rtol=0, atol=0)
