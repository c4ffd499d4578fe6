"""
Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on ROCm).

The reference has no distributed layer at all; what shards is the data:

  * group_by-partitioned regressions (the headline path) shard BY GROUP KEY: every rank owns a contiguous
    range of groups and runs the grouped kernels on it.  There is no data-path collective; only the
    results (n_groups x p' coefficients + null flags) are gathered when a caller wants them on one rank.
  * a single big regression shards BY ROW RANGE: every rank builds the moment matrix of its rows and ONE
    all-reduce(SUM) of that (p+2)^2 block (2.6 KB at p = 16 -- latency bound, any algorithm) makes the
    normal equations global; the O(p^3) solve is replicated.  lin_reg_report adds one more all-reduce of
    (sum e^2, sum w e^2) after the residual pass.
  * rolling regressions shard BY ROW RANGE with a halo: rank r also reads the window-1 rows in front of its range
    (its windows reach back into them) and drops their outputs; no collective.  Expanding ("recursive")
    regressions need what came before: every rank builds the moment matrix of its rows, ONE all-gather of
    those (p+2)^2 blocks gives each rank the exclusive prefix, which seeds its local expanding fit
    (pds_recursive_lr_seeded_*).
  * coordinate-descent sweeps do not shard (sequential Gauss-Seidel on a p x p matrix): replicas only.

The compute steps are injected (`moments_fn`, `solve_fn`, `grouped_fn`, `rolling_fn`, `recursive_fn`) so the orchestration can be
exercised on CPU with the gloo backend (tests/test_parallel_gloo.py); by default they are the HIP path.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def shard_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` (first n_items % world ranks get one more)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_groups_by_rows(group_offsets: np.ndarray, world: int) -> list[tuple[int, int]]:
    """
    Range-partition groups (given by n_groups+1 row offsets) over `world` ranks so that every rank gets
    about the same number of ROWS (the Gram build is bandwidth bound in rows, not in groups).
    Returns [(g_lo, g_hi)] per rank; consecutive and covering, possibly empty for tiny inputs.
    """
    off = np.asarray(group_offsets, dtype=np.int64)
    n_groups = len(off) - 1
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world):
        target = off[0] + (total * r) // world
        g = int(np.searchsorted(off, target, side="left"))
        cuts.append(min(max(g, cuts[-1]), n_groups))
    cuts.append(n_groups)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _dist():
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
    return dist


def _hip_moments(xs, y, weights=None):
    from . import lstsq

    return lstsq.gram_moments(*xs, target=y, weights=weights, out_device=True)


def _hip_solve(moments, **kw):
    from . import lstsq

    return lstsq.lin_reg_from_moments(moments, **kw)


def _hip_grouped(xs, y, offsets, **kw):
    from . import lstsq

    return lstsq.lin_reg_by(*xs, target=y, group_offsets=offsets, **kw)


def lin_reg_row_sharded(xs_local: Sequence, y_local, *, weights_local=None, moments_fn: Callable | None = None,
                        solve_fn: Callable | None = None, group=None, **lin_reg_kwargs):
    """
    pds.lin_reg over a frame whose rows are sharded across ranks.  Every rank passes its own row range of
    every column; every rank returns the same coefficient vector (or None when the gate fires).
    """
    import torch

    dist = _dist()
    moments_fn = moments_fn or _hip_moments
    solve_fn = solve_fn or _hip_solve
    m = moments_fn(xs_local, y_local, weights_local)
    if not isinstance(m, torch.Tensor):
        m = torch.as_tensor(np.asarray(m))
    m = m.contiguous()
    dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)  # the one exchange step of the path
    return solve_fn(m, **lin_reg_kwargs)


def lin_reg_by_group_sharded(xs, y, group_offsets, *, rank: int | None = None, world: int | None = None,
                             grouped_fn: Callable | None = None, gather_to: int | None = None, group=None,
                             **lin_reg_kwargs):
    """
    group_by(key).agg(pds.lin_reg(...)) sharded by group key.  `xs`, `y`, `group_offsets` describe the whole
    frame as visible to this rank (e.g. a host-resident Arrow table every rank can slice, or each rank's
    own copy); the rank extracts ITS groups' rows, runs the grouped kernels, and returns
    (g_lo, g_hi, coeffs_local, is_null_local).  With gather_to = r the per-rank results are gathered on
    rank r, which additionally returns the assembled (coeffs, is_null) for all groups.
    """
    import torch

    dist = _dist()
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    grouped_fn = grouped_fn or _hip_grouped
    off = np.asarray(group_offsets.cpu() if hasattr(group_offsets, "cpu") else group_offsets, dtype=np.int64)
    parts = shard_groups_by_rows(off, world)
    g_lo, g_hi = parts[rank]
    r_lo, r_hi = int(off[g_lo]), int(off[g_hi])
    loc_off = off[g_lo : g_hi + 1] - r_lo
    xs_loc = [x[r_lo:r_hi] for x in xs]
    y_loc = y[r_lo:r_hi]
    if g_hi > g_lo:
        co, nu = grouped_fn(xs_loc, y_loc, loc_off, **lin_reg_kwargs)
    else:
        co, nu = None, None
    if gather_to is None:
        return g_lo, g_hi, co, nu
    # results only: n_groups_r x p' values + flags per rank (8 MB per rank at config 3) -> gather
    pp = len(xs) + int(bool(lin_reg_kwargs.get("add_bias", False)))
    def to_t(a, shape, dtype):
        if a is None:
            return torch.zeros(shape, dtype=dtype)
        return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    co_t = to_t(co, (0, pp), torch.float64)
    nu_t = to_t(nu, (0,), torch.uint8).to(torch.uint8)
    sizes = [parts[r][1] - parts[r][0] for r in range(world)]
    dev = co_t.device
    if rank == gather_to:
        co_list = [torch.empty((sizes[r], pp), dtype=co_t.dtype, device=dev) for r in range(world)]
        nu_list = [torch.empty((sizes[r],), dtype=torch.uint8, device=dev) for r in range(world)]
    else:
        co_list = nu_list = None
    # gather with unequal sizes: point-to-point to the root (every peer uses its own xGMI link to the root)
    if rank == gather_to:
        co_list[rank].copy_(co_t)
        nu_list[rank].copy_(nu_t)
        for r in range(world):
            if r != rank and sizes[r] > 0:
                dist.recv(co_list[r], src=r, group=group)
                dist.recv(nu_list[r], src=r, group=group)
        return g_lo, g_hi, torch.cat(co_list, 0), torch.cat(nu_list, 0)
    if sizes[rank] > 0:
        dist.send(co_t.contiguous(), dst=gather_to, group=group)
        dist.send(nu_t.contiguous(), dst=gather_to, group=group)
    return g_lo, g_hi, co, nu


def _hip_rolling(xs, y, **kw):
    from . import lstsq

    return lstsq.rolling_lin_reg(*xs, target=y, **kw)


def _hip_recursive(xs, y, **kw):
    from . import lstsq

    return lstsq.recursive_lin_reg(*xs, target=y, **kw)


def rolling_lin_reg_row_sharded(xs, y, window_size: int, *, rank: int | None = None, world: int | None = None,
                                rolling_fn: Callable | None = None, group=None, **rolling_kwargs):
    """
    pds.rolling_lin_reg over a frame every rank can slice (`xs`, `y`: the whole columns as visible to this rank).
    Rank r owns output rows [lo, hi) = shard_bounds(N, world, r) and runs the rolling kernel on rows
    [lo - (window_size - 1), hi): the halo rows only feed the first windows, their outputs are dropped.  Returns
    (lo, hi, coeffs_local, pred_local, valid_local); rows < window_size - 1 of the frame stay invalid on rank 0.
    No collective on the data path (outputs stay sharded like the inputs).
    """
    dist = _dist()
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    rolling_fn = rolling_fn or _hip_rolling
    n = len(y)
    lo, hi = shard_bounds(n, world, rank)
    if hi <= lo:
        return lo, hi, None, None, None
    h0 = max(0, lo - (int(window_size) - 1))
    if hi - h0 < window_size:  # (only a frame shorter than one window) nothing valid here
        return lo, hi, None, None, None
    co, pr, va = rolling_fn([x[h0:hi] for x in xs], y[h0:hi], window_size=window_size, **rolling_kwargs)
    k = lo - h0
    return lo, hi, co[k:], pr[k:], va[k:]


def recursive_lin_reg_row_sharded(xs_local: Sequence, y_local, start_with: int, *, moments_fn: Callable | None = None,
                                  recursive_fn: Callable | None = None, group=None, **recursive_kwargs):
    """
    pds.recursive_lin_reg over row-sharded columns (rank r holds the r-th contiguous row range).  One all-gather of
    the per-rank moment matrices; rank r seeds its expanding fit with the sum of the matrices of ranks < r.
    Returns this rank's (coeffs, pred, valid).
    """
    import torch

    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    moments_fn = moments_fn or _hip_moments
    recursive_fn = recursive_fn or _hip_recursive
    m = moments_fn(xs_local, y_local, None)
    if not isinstance(m, torch.Tensor):
        m = torch.as_tensor(np.asarray(m))
    m = m.contiguous()
    blocks = [torch.empty_like(m) for _ in range(world)]
    dist.all_gather(blocks, m, group=group)  # the one exchange step: world x (p+2)^2 values
    seed = None
    if rank > 0:
        seed = torch.stack(blocks[:rank]).sum(dim=0)  # fixed order: identical on every run
    return recursive_fn(xs_local, y_local, start_with=start_with, seed_moments=seed, **recursive_kwargs)
