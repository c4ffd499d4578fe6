"""
Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on ROCm).

The reference has no distributed layer at all; what shards is the data:

  * group_by-partitioned regressions (the headline path) shard BY GROUP KEY: every rank owns a contiguous
    range of groups and runs the grouped kernels on it.  A frame that lives on one rank is SCATTERED first (grouped
    point-to-point sends, root -> every peer over its own xGMI link); the results (n_groups x p' coefficients + null
    flags) are GATHERED to the requesting rank the same way, piece by piece while the next piece is computed.
  * a single big regression shards BY ROW RANGE: every rank builds the moment matrix of its rows and ONE
    all-reduce(SUM) of that (p+2)^2 block (2.6 KB at p = 16 -- latency bound, any algorithm) makes the
    normal equations global; the O(p^3) solve is replicated.  lin_reg_report adds one more all-reduce of
    [sum e^2 | sum w e^2 | HC meat] after the residual pass (lin_reg_report_row_sharded).
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


def _collective_device(dist, group, fallback):
    """Where a small control tensor of a collective must live: the current GPU under the nccl (= RCCL) backend, else `fallback`."""
    import torch

    try:
        backend = str(dist.get_backend(group))
    except Exception:
        backend = ""
    if "nccl" in backend and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return fallback


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


def _hip_grouped_out(xs, y, offsets, **kw):
    from . import lstsq

    return lstsq.lin_reg_by(*xs, target=y, group_offsets=offsets, **kw)


def chunk_bounds(n_groups: int, chunks: int) -> list[tuple[int, int]]:
    """`chunks` consecutive ranges of equal group COUNT (a function of the count alone: the gathering rank can reproduce it)."""
    chunks = max(1, min(int(chunks), max(int(n_groups), 1)))
    return [shard_bounds(n_groups, chunks, c) for c in range(chunks)]


def auto_chunks(world: int, result_bytes_per_peer: int, shard_input_bytes: int, *, link_GBps: float = 45.0,
                stream_GBps: float = 5500.0, launch_us: float = 25.0) -> int:
    """
    How many pieces a rank's shard is fitted in so that a finished piece's results travel while the next piece is computed.
    With kernel time T_k = shard bytes / stream rate and transfer time T_x = result bytes / link rate (one xGMI link per peer into
    the gathering rank), a c-piece pipeline takes max(T_k, T_x) + min(T_k, T_x) / c + c * (one more launch of each): the minimum is at
    c = sqrt(min(T_k, T_x) / launch).  1e6 groups x 100 rows x 16 f64 features over 8 ranks: T_k = 0.31 ms, T_x = 0.36 ms -> 4 pieces;
    a world of one has nothing to overlap -> 1.  The rates are defaults for the model, not measurements of the run.
    """
    if world <= 1 or result_bytes_per_peer <= 0 or shard_input_bytes <= 0:
        return 1
    t_x = result_bytes_per_peer / (link_GBps * 1e9)
    t_k = shard_input_bytes / (stream_GBps * 1e9)
    return int(max(1, min(8, round((min(t_x, t_k) / (launch_us * 1e-6)) ** 0.5))))


class GroupedShardPlan:
    """
    The compute + gather leg of the group-sharded regression on a rank that already HOLDS its shard (rows of groups
    parts[rank] = [g_lo, g_hi), `loc_off` rebased to the shard), prepared ONCE and then run step after step with nothing but the
    launches in a step:

      * the result buffers are persistent: the gathering rank owns the assembled [n_groups, p'] coefficients + [n_groups] null flags,
        its own shard is fitted IN PLACE into rows [g_lo, g_hi) of them (the kernel gets that address: no copy), and its local
        results are views of those rows; a peer owns its shard's results;
      * the shard is fitted in `chunks` pieces (default: auto_chunks); with the library's own fit every piece is a prepared call
        (lstsq.GroupedFit: pointer table, offsets, parameters set up here) -- one C call per piece and step;
      * a peer hands every finished piece to the gathering rank as ONE grouped point-to-point launch (coefficients + flags) on
        RCCL's stream while its next piece is computed; the gathering rank posts, before its own fit, one grouped receive per piece
        index straight into the rows of the assembled result -- every peer's traffic rides its own xGMI link, no staging copy.

    Building a plan is COLLECTIVE when `chunks` is left to the model (every rank of the group builds its plan at the same point: one
    all-reduce agrees on the piece count); with an explicit `chunks` every rank must pass the same value.
    `step()` returns (coeffs_local, is_null_local) and, on the gathering rank, additionally (coeffs, is_null) of all groups: views of
    the persistent buffers, valid until the next step().  The dtype on the wire is `result_dtype`; by default what the library's
    grouped fit returns (the config dtype) or, with an injected `grouped_fn`, the targets' dtype.

    `direct=True` (device-resident shards, the library's own fit; opt-in until it has run across two devices): NO gather step at all.
    The gathering rank allocates the assembled result as an exportable block (lstsq.DeviceBlock), every peer maps it (IPC handle,
    exchanged once when the plan is built) and its prepared fit gets `mapped + its rows' offset` as output address: the kernel's
    coefficient / flag stores cross the peer's xGMI link while it runs (16 MB per peer inside a 0.31 ms kernel on the headline frame
    over 8 ranks = 52 GB/s per link), one piece per rank, no send / receive launches.  Completion is a word per rank in a second
    shared block: a peer posts the step number behind its fit (`Context.signal_post`: a one-lane kernel, stream ordered), the
    gathering rank's stream waits for all of them (`Context.signal_wait`: a one-wave polling kernel) -- no collective, no host
    synchronisation; consecutive steps queue back to back.  Peers return (None, None) from step() (their results live on the
    gathering rank only); `chunks` / `chunk_model` do not apply.  `close()` (collective) unmaps the blocks before the owner frees them.

    `chunk_model`: overrides of auto_chunks' rates ({"link_GBps", "stream_GBps", "launch_us"}) when `chunks` is left to the model.
    """

    def __init__(self, xs_loc, y_loc, loc_off, parts, *, rank: int, gather_to: int | None = 0, chunks: int | None = None,
                 grouped_fn: Callable | None = None, group=None, result_dtype=None, ctx=None, direct: bool = False,
                 chunk_model: dict | None = None, **lin_reg_kwargs):
        import torch

        self._dist = dist = _dist()
        self.rank, self.parts, self.root, self.group = rank, parts, gather_to, group
        world = len(parts)
        g_lo, g_hi = parts[rank]
        ng = g_hi - g_lo
        pp = len(xs_loc) + int(bool(lin_reg_kwargs.get("add_bias", False)))
        is_t = isinstance(y_loc, torch.Tensor)  # (NumPy 2 arrays have a .device too)
        dev = y_loc.device if is_t else torch.device("cpu")
        default_fn = grouped_fn is None
        if result_dtype is not None:
            cdt = result_dtype
        elif default_fn:
            from . import lstsq

            cdt = torch.float64 if lstsq._dtype() == np.float64 else torch.float32
        else:
            cdt = y_loc.dtype if is_t else torch.from_numpy(np.asarray(y_loc)[:0]).dtype
        off_h = np.asarray(loc_off.cpu() if hasattr(loc_off, "cpu") else loc_off, dtype=np.int64)  # (row bounds of the pieces: host)
        is_root = gather_to is not None and rank == gather_to
        total = parts[-1][1]
        self.direct = bool(direct) and gather_to is not None
        if self.direct:
            if not (default_fn and is_t and y_loc.is_cuda):
                raise ValueError("direct=True: device-resident shards fitted by the library (no injected grouped_fn)")
            self._init_direct(xs_loc, y_loc, loc_off, ng, pp, cdt, dev, is_root, g_lo, total, world, ctx, lin_reg_kwargs)
            return
        if chunks is None:
            esz = torch.empty((), dtype=cdt).element_size()
            in_sz = y_loc.element_size() if is_t else np.asarray(y_loc).dtype.itemsize
            peers = [hi - lo for r, (lo, hi) in enumerate(parts) if r != gather_to]
            chunks = auto_chunks(world if gather_to is not None else 1, max(peers, default=0) * (pp * esz + 1),
                                 int(off_h[-1] - off_h[0]) * (len(xs_loc) + 1) * in_sz, **(chunk_model or {}))
            if world > 1 and gather_to is not None:
                # the piece count decides how the gathering rank cuts every peer's rows: ranks whose shards differ by a group must not
                # round the model to different counts -- one tiny all-reduce(MAX) when the plan is built (building a plan is collective)
                # (on the backend's device: RCCL reduces device tensors only, whatever space the shard's columns live in)
                agree = torch.tensor([int(chunks)], dtype=torch.int64, device=_collective_device(dist, group, dev))
                dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=group)
                chunks = int(agree.item())
        self.chunks = chunks = max(1, int(chunks))
        # ---- persistent results
        if is_root:
            self.co_all = torch.empty((total, pp), dtype=cdt, device=dev)
            self.nu_all = torch.empty((total,), dtype=torch.uint8, device=dev)
            self.co_loc, self.nu_loc = self.co_all[g_lo:g_hi], self.nu_all[g_lo:g_hi]
        else:
            self.co_all = self.nu_all = None
            self.co_loc = torch.empty((ng, pp), dtype=cdt, device=dev)
            self.nu_loc = torch.empty((ng,), dtype=torch.uint8, device=dev)
        # ---- the receives of the gathering rank: one grouped launch per piece index (what the peers send, in the order they send it)
        self._recv_batches = []
        if is_root:
            per_peer = {r: [b for b in chunk_bounds(parts[r][1] - parts[r][0], chunks) if b[1] > b[0]] for r in range(world) if r != rank}
            for c in range(chunks):
                ops = []
                for r, bl in per_peer.items():
                    if c < len(bl):
                        lo, hi = parts[r][0] + bl[c][0], parts[r][0] + bl[c][1]
                        ops.append(dist.P2POp(dist.irecv, self.co_all[lo:hi], r, group))
                        ops.append(dist.P2POp(dist.irecv, self.nu_all[lo:hi], r, group))
                if ops:
                    self._recv_batches.append(ops)
        # ---- this rank's pieces: a prepared fit (or the injected function + a copy into place) and, on a peer, its send
        self._runs, self._send_batches = [], []
        use_prepared = False
        if default_fn and is_t and y_loc.is_cuda:  # (the wire dtype is the library's result dtype: the fit writes into place)
            import inspect

            from . import lstsq

            # kwargs that only lin_reg_by takes (null_policy, weights): the unprepared call, as before the plan existed
            prepared_kw = set(inspect.signature(lstsq.GroupedFit.__init__).parameters)
            use_prepared = cdt == (torch.float64 if lstsq._dtype() == np.float64 else torch.float32) and set(lin_reg_kwargs) <= prepared_kw
        # the library's own fit runs on the caller's context on either path; an injected function gets exactly the caller's kwargs
        call_kw = dict(lin_reg_kwargs, ctx=ctx) if (default_fn and ctx is not None) else lin_reg_kwargs
        for c_lo, c_hi in (b for b in chunk_bounds(ng, chunks) if ng > 0 and b[1] > b[0]):
            r0, r1 = int(off_h[c_lo]), int(off_h[c_hi])
            co_v, nu_v = self.co_loc[c_lo:c_hi], self.nu_loc[c_lo:c_hi]
            xs_p, y_p = [x[r0:r1] for x in xs_loc], y_loc[r0:r1]
            sub_off = loc_off[c_lo: c_hi + 1] - r0
            if use_prepared:
                fit = lstsq.GroupedFit(*xs_p, target=y_p, group_offsets=sub_off, out=co_v, out_null=nu_v, ctx=ctx, **lin_reg_kwargs)
                self._runs.append(fit.run)
            else:
                self._runs.append(self._injected(grouped_fn or _hip_grouped_out, xs_p, y_p, sub_off, co_v, nu_v, call_kw))
            if gather_to is not None and not is_root:
                self._send_batches.append([dist.P2POp(dist.isend, co_v, gather_to, group), dist.P2POp(dist.isend, nu_v, gather_to, group)])

    def _init_direct(self, xs_loc, y_loc, loc_off, ng, pp, cdt, dev, is_root, g_lo, total, world, ctx, lin_reg_kwargs):
        """The direct gather: one exportable result block on the gathering rank, mapped by every peer; one prepared fit per rank."""
        import torch

        from . import lstsq

        dist = self._dist
        if cdt != (torch.float64 if lstsq._dtype() == np.float64 else torch.float32):
            raise ValueError("direct=True writes the library's result dtype")
        esz = torch.empty((), dtype=cdt).element_size()
        co_bytes = (total * pp * esz + 255) & ~255  # [coefficients | flags] in ONE allocation: one handle
        self.chunks = 1
        dev_i = dev.index if dev.index is not None else torch.cuda.current_device()
        self._ctx = ctx or lstsq.default_context()
        meta = [None]
        if is_root:
            self._block = lstsq.DeviceBlock(co_bytes + total, dev_i)
            self._sig = lstsq.DeviceBlock(4096, dev_i)
            self._sig.tensor(torch.uint8, (4096,)).zero_()
            torch.cuda.synchronize(dev)
            self.co_all = self._block.tensor(cdt, (total, pp))
            self.nu_all = self._block.tensor(torch.uint8, (total,), co_bytes)
            meta = [(self._block.handle(), self._sig.handle())]
        dist.broadcast_object_list(meta, src=self.root, group=self.group)  # (collective: every rank builds its plan at the same point)
        if is_root:
            self.co_loc, self.nu_loc = self.co_all[g_lo:g_lo + ng], self.nu_all[g_lo:g_lo + ng]
        else:
            self._block = lstsq.DeviceBlock.open(meta[0][0], dev_i, co_bytes + total)
            self._sig = lstsq.DeviceBlock.open(meta[0][1], dev_i, 4096)
            self.co_all = self.nu_all = self.co_loc = self.nu_loc = None
        base = self._block.addr
        self._runs, self._send_batches, self._recv_batches = [], [], []
        if ng > 0:
            fit = lstsq.GroupedFit(*xs_loc, target=y_loc, group_offsets=loc_off[0: ng + 1], out_addr=base + g_lo * pp * esz,
                                   out_null_addr=base + co_bytes + g_lo, ctx=self._ctx, **lin_reg_kwargs)
            self._runs.append(fit.run)
        # completion words: slot k of the signal block belongs to the k-th peer (ranks other than the gathering one, in rank order)
        peers = [r for r in range(world) if r != self.root]
        self._n_peers = len(peers)
        self._my_word = None if is_root else self._sig.addr + 4 * peers.index(self.rank)
        self._seq = 0
        self._torch_dev = dev

    def close(self) -> None:
        """Direct plans: peers unmap the shared blocks, then the owner frees them (collective).  Other plans: nothing to do."""
        if not self.direct or getattr(self, "_block", None) is None:
            return
        import torch

        torch.cuda.synchronize(self._torch_dev)
        self._runs = []
        if self.co_all is None:
            self._block.close()
            self._sig.close()
        self._dist.barrier(group=self.group)
        if self.co_all is not None:
            self.co_all = self.nu_all = self.co_loc = self.nu_loc = None  # (views of the block)
            self._block.close()
            self._sig.close()
        self._block = self._sig = None

    @staticmethod
    def _injected(fn, xs_p, y_p, sub_off, co_v, nu_v, kw):
        import torch

        def run():
            co, nu = fn(xs_p, y_p, sub_off, **kw)
            co = co if isinstance(co, torch.Tensor) else torch.as_tensor(np.asarray(co))
            nu = nu if isinstance(nu, torch.Tensor) else torch.as_tensor(np.asarray(nu))
            co_v.copy_(co)  # (casts to the wire dtype / moves to the result's device)
            nu_v.copy_(nu)

        return run

    def _step_direct(self):
        self._seq += 1
        if not self._runs:  # (an empty shard still follows the stream its words are posted on)
            self._ctx.follow_torch_stream(self._torch_dev)
        for run in self._runs:
            run()
        if self.co_all is not None:
            if self._n_peers:
                self._ctx.signal_wait(self._sig.addr, self._n_peers, self._seq)
            return self.co_loc, self.nu_loc, self.co_all, self.nu_all
        self._ctx.signal_post(self._my_word, self._seq)
        return None, None

    def step(self):
        if self.direct:
            return self._step_direct()
        batch = self._dist.batch_isend_irecv
        reqs = []
        for ops in self._recv_batches:
            reqs += batch(ops)
        if self._send_batches:
            for run, ops in zip(self._runs, self._send_batches):
                run()
                reqs += batch(ops)
        else:
            for run in self._runs:
                run()
        for q in reqs:
            q.wait()
        if self.co_all is not None:
            return self.co_loc, self.nu_loc, self.co_all, self.nu_all
        return self.co_loc, self.nu_loc


def lin_reg_by_group_local_shard(xs_loc, y_loc, loc_off, parts, *, rank: int, gather_to: int | None = 0, chunks: int | None = 1,
                                 grouped_fn: Callable | None = None, group=None, result_dtype=None, **lin_reg_kwargs):
    """
    One step of a GroupedShardPlan built for this call (a caller that repeats the step keeps the plan instead).
    Returns (coeffs_local, is_null_local) and, on the gathering rank, additionally the assembled (coeffs, is_null).
    """
    return GroupedShardPlan(xs_loc, y_loc, loc_off, parts, rank=rank, gather_to=gather_to, chunks=chunks, grouped_fn=grouped_fn,
                            group=group, result_dtype=result_dtype, **lin_reg_kwargs).step()


def scatter_frame_by_groups(xs, y, group_offsets, *, root: int = 0, group=None, device=None):
    """
    The scatter leg for a frame that is resident on ONE rank ("device-resident frame on GPU0: RCCL scatter"):
    `root` passes the whole columns and the n_groups + 1 row offsets, every other rank passes None.  The groups are range-
    partitioned balanced in rows; the root sends every peer the row range of every column (views of the resident columns: no
    staging copy) plus the peer's rebased offsets as ONE grouped point-to-point launch -- root -> 7 peers uses all 7 xGMI
    links at once, 1/8 of the frame per link -- and every peer receives straight into fresh tensors.
    Returns (xs_local, y_local, offsets_local (int64, rebased to the shard), parts) on every rank.
    """
    import torch

    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [None]
    if rank == root:
        off = np.asarray(group_offsets.cpu() if hasattr(group_offsets, "cpu") else group_offsets, dtype=np.int64)
        parts = shard_groups_by_rows(off, world)
        rows = [(int(off[lo]), int(off[hi])) for lo, hi in parts]
        meta = [dict(parts=parts, rows=rows, n_cols=len(xs), dtype=str(y.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(meta, src=root, group=group)
    m = meta[0]
    parts, rows, nc = m["parts"], m["rows"], m["n_cols"]
    dt = getattr(torch, m["dtype"])
    if rank == root:
        dev = y.device
        off_t = torch.as_tensor(off, device=dev)
        ops, keep = [], []
        for r in range(world):
            if r == root or parts[r][1] <= parts[r][0]:
                continue
            r0, r1 = rows[r]
            for col in (*xs, y):
                ops.append(dist.P2POp(dist.isend, col[r0:r1], r, group))
            lo, hi = parts[r]
            o = (off_t[lo: hi + 1] - r0).contiguous()
            keep.append(o)
            ops.append(dist.P2POp(dist.isend, o, r, group))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        r0, r1 = rows[root]
        lo, hi = parts[root]
        return [x[r0:r1] for x in xs], y[r0:r1], off_t[lo: hi + 1] - r0, parts
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    r0, r1 = rows[rank]
    lo, hi = parts[rank]
    n = r1 - r0
    xs_l = [torch.empty(n, dtype=dt, device=dev) for _ in range(nc)]
    y_l = torch.empty(n, dtype=dt, device=dev)
    off_l = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
    if hi > lo:
        ops = [dist.P2POp(dist.irecv, t, root, group) for t in (*xs_l, y_l, off_l)]
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return xs_l, y_l, off_l, parts


def lin_reg_by_group_sharded(xs, y, group_offsets, *, rank: int | None = None, world: int | None = None,
                             grouped_fn: Callable | None = None, gather_to: int | None = None, chunks: int = 1, group=None,
                             **lin_reg_kwargs):
    """
    group_by(key).agg(pds.lin_reg(...)) sharded by group key.  `xs`, `y`, `group_offsets` describe the whole
    frame as visible to this rank (e.g. a host-resident Arrow table every rank can slice, or each rank's
    own copy); the rank extracts ITS groups' rows, runs the grouped kernels, and returns
    (g_lo, g_hi, coeffs_local, is_null_local).  With gather_to = r the per-rank results are gathered on
    rank r (lin_reg_by_group_local_shard), which additionally returns the assembled (coeffs, is_null) for all groups.
    A frame resident on one rank only goes through scatter_frame_by_groups first.
    """
    dist = _dist()
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    off = np.asarray(group_offsets.cpu() if hasattr(group_offsets, "cpu") else group_offsets, dtype=np.int64)
    parts = shard_groups_by_rows(off, world)
    g_lo, g_hi = parts[rank]
    r_lo, r_hi = int(off[g_lo]), int(off[g_hi])
    loc_off = off[g_lo: g_hi + 1] - r_lo
    res = lin_reg_by_group_local_shard([x[r_lo:r_hi] for x in xs], y[r_lo:r_hi], loc_off, parts, rank=rank, gather_to=gather_to,
                                       chunks=chunks, grouped_fn=grouped_fn, group=group, **lin_reg_kwargs)
    if gather_to is not None and rank == gather_to:
        return g_lo, g_hi, res[2], res[3]
    co, nu = res[0], res[1]
    if g_hi <= g_lo:
        co, nu = None, None
    return g_lo, g_hi, co, nu


def _hip_report_fit(moments, **kw):
    from . import lstsq

    return lstsq.report_fit_from_moments(moments, **kw)


def _hip_report_partials(xs, y, weights, beta, inv, **kw):
    from . import lstsq

    return lstsq.report_partials(*xs, target=y, weights=weights, beta=beta, inv=inv, **kw)


def _hip_report_finish(beta, inv, partials, **kw):
    from . import lstsq

    return lstsq.report_finish(beta, inv, partials, **kw)


def lin_reg_report_row_sharded(xs_local: Sequence, y_local, *, add_bias: bool = False, std_err: str = "se", weights_local=None,
                               y_var: float | None = None, moments_fn: Callable | None = None, fit_fn: Callable | None = None,
                               partials_fn: Callable | None = None, finish_fn: Callable | None = None, group=None):
    """
    pds.lin_reg_report (and the WLS report with `weights_local`) over row-sharded columns: the reference's arithmetic
    (linear_regression.rs:854-909) with its two passes over X turned into two exchange steps --
      1. local moment block -> all-reduce(SUM)                     ((p+2)^2 values: 2.6 KB at p = 16)
      2. (X'X)^-1 and beta from the summed block, replicated on every rank (deterministic: no broadcast needed)
      3. local residual pass with that beta / inverse: [sum e^2 | sum w e^2 | meat] -> all-reduce(SUM)   (2 + (p+2)^2 values)
      4. the O(p'^2) epilogue (SE / HC0-3, t, p, CI, r2), replicated.
    `y_var` defaults to the ddof = 1 variance of the whole target, taken from the summed moments.  Every rank returns the report.
    """
    import torch

    dist = _dist()
    moments_fn = moments_fn or _hip_moments
    fit_fn = fit_fn or _hip_report_fit
    partials_fn = partials_fn or _hip_report_partials
    finish_fn = finish_fn or _hip_report_finish
    m = moments_fn(xs_local, y_local, weights_local)
    if not isinstance(m, torch.Tensor):
        m = torch.as_tensor(np.asarray(m))
    m = m.contiguous()
    dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)  # exchange step 1
    n_local = torch.tensor([len(y_local)], dtype=torch.int64, device=m.device)
    dist.all_reduce(n_local, op=dist.ReduceOp.SUM, group=group)  # (row count: the weighted block carries sum(w) instead of n)
    n_total = int(n_local.item())
    mh = m.detach().cpu().numpy()
    q = mh.shape[0]
    if y_var is None:
        if weights_local is not None:
            raise ValueError("y_var (the unweighted ddof = 1 variance of the whole target) must be given for the weighted report")
        sy, syy = float(mh[q - 2, q - 1]), float(mh[q - 1, q - 1])
        y_var = (syy - sy * sy / n_total) / (n_total - 1.0)
    beta, inv = fit_fn(mh, add_bias=add_bias)
    part = partials_fn(xs_local, y_local, weights_local, beta, inv, add_bias=add_bias, std_err=std_err)
    pt = torch.as_tensor(np.asarray(part, dtype=np.float64)).to(m.device)
    dist.all_reduce(pt, op=dist.ReduceOp.SUM, group=group)  # exchange step 2
    return finish_fn(beta, inv, pt.cpu().numpy(), n_rows_total=n_total, y_var=y_var, add_bias=add_bias,
                     weighted=weights_local is not None, std_err=std_err)


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
