"""
Synthetic frames of SURVEY.md 8(d) ("Synthetic inputs"), generated on the GPU with seeded torch generators -- shared by
bench.py and the BASELINE-size parity tests so that both work on the same data.  Column-major: one contiguous tensor per column.
"""
from __future__ import annotations

import numpy as np
import torch


def c3_frame(n_groups: int = 1_000_000, p: int = 8, seed: int = 2, device="cuda", collinear_frac: float = 1e-3,
             fixed_size: int | None = None, dtype=torch.float64):
    """
    C3: G groups, rows per group ~ Poisson(100) clipped to [16, 256] (or `fixed_size`), p features x ~ N(0,1), per-group
    beta ~ N(0,1), noise 0.1, `collinear_frac` of the groups made collinear (x2 = 2 x1) to exercise the rank gate.
    Returns dict(xs=[p tensors], y, offsets (G+1 int64, device), sizes, keys (N int64, sorted), beta (G x p), collinear (G bool)).
    """
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if fixed_size is None:
        lam = torch.full((n_groups,), 100.0, device=device)
        sizes = torch.poisson(lam, generator=g).clamp_(16, 256).to(torch.int64)
    else:
        sizes = torch.full((n_groups,), int(fixed_size), device=device, dtype=torch.int64)
    offsets = torch.zeros(n_groups + 1, dtype=torch.int64, device=device)
    torch.cumsum(sizes, 0, out=offsets[1:])
    n = int(offsets[-1].item())
    gid = torch.repeat_interleave(torch.arange(n_groups, device=device), sizes)  # = the sorted int64 key column
    beta = torch.randn(n_groups, p, dtype=torch.float64, device=device, generator=g)
    collinear = torch.zeros(n_groups, dtype=torch.bool, device=device)
    if p >= 2 and collinear_frac > 0:
        k = max(1, int(round(n_groups * collinear_frac)))
        collinear[torch.randperm(n_groups, device=device, generator=g)[:k]] = True
    xs = []
    y = 0.1 * torch.randn(n, dtype=torch.float64, device=device, generator=g)
    for j in range(p):
        x = torch.randn(n, dtype=torch.float64, device=device, generator=g)
        if j == 1:
            rows = collinear[gid]
            x = torch.where(rows, 2.0 * xs[0], x)
        xs.append(x)
        y += x * beta[:, j][gid]
    if dtype != torch.float64:
        xs = [x.to(dtype) for x in xs]
        y = y.to(dtype)
    return dict(xs=xs, y=y, offsets=offsets, sizes=sizes, keys=gid, beta=beta, collinear=collinear, n_rows=n)


def c2_frame(n: int = 100_000_000, p: int = 16, seed: int = 1, device="cuda"):
    """C2: x ~ U(0,1), beta_j = (-1)^j (0.05 + 0.03 j) except beta_3 = beta_11 = 0, y = X beta + 1e-2 N(0,1)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    beta = np.array([(-1.0) ** j * (0.05 + 0.03 * j) for j in range(p)])
    for j in (3, 11):
        if j < p:
            beta[j] = 0.0
    xs = [torch.rand(n, dtype=torch.float64, device=device, generator=g) for _ in range(p)]
    y = 1e-2 * torch.randn(n, dtype=torch.float64, device=device, generator=g)
    for j in range(p):
        if beta[j] != 0.0:
            y.add_(xs[j], alpha=float(beta[j]))
    return dict(xs=xs, y=y, beta=beta)


def c4_frame(n: int = 100_000_000, p: int = 8, seed: int = 3, device="cuda"):
    """C4: x ~ U(0,1), y = X beta + 1e-3 N(0,1)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    beta = np.array([(-1.0) ** j * (0.3 + 0.1 * j) for j in range(p)])
    xs = [torch.rand(n, dtype=torch.float64, device=device, generator=g) for _ in range(p)]
    y = 1e-3 * torch.randn(n, dtype=torch.float64, device=device, generator=g)
    for j in range(p):
        y.add_(xs[j], alpha=float(beta[j]))
    return dict(xs=xs, y=y, beta=beta)


def c5_frame(n: int = 10_000_000, p: int = 512, seed: int = 4, device="cuda", block: int = 1_000_000):
    """
    C5: f32, x ~ N(0,1) with AR(0.5) column correlation, 32 non-zero true coefficients, noise 0.5.  Returns dict(X (p x n,
    row j = column j of the frame, contiguous), y, beta).  Generated in row blocks (the n x p noise matrix of one block at a time).
    """
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    beta = np.zeros(p)
    beta[rng.choice(p, 32, replace=False)] = rng.normal(size=32)
    bt = torch.from_numpy(beta).to(device)
    X = torch.empty(p, n, dtype=torch.float32, device=device)
    y = torch.empty(n, dtype=torch.float32, device=device)
    c = float(np.sqrt(0.75))
    for r0 in range(0, n, block):
        r1 = min(n, r0 + block)
        E = torch.randn(p, r1 - r0, dtype=torch.float32, device=device, generator=g)
        Xb = X[:, r0:r1]
        Xb[0] = E[0]
        for j in range(1, p):
            torch.add(E[j] * c, Xb[j - 1], alpha=0.5, out=Xb[j])
        y[r0:r1] = (bt.float() @ Xb) + 0.5 * torch.randn(r1 - r0, dtype=torch.float32, device=device, generator=g)
        del E
    return dict(X=X, y=y, beta=beta)


def headline_frame(n_groups: int = 1_000_000, rows_per_group: int = 100, p: int = 16, seed: int = 1234, device="cuda"):
    """
    The frame bench.py's `value` is measured on (BASELINE.json's metric: grouped lstsq, 1e8 rows x 16 f64 features):
    x ~ N(0,1), per-group beta ~ N(0,1), noise 0.1, a fixed number of rows per group.  Returns (xs, y).
    """
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    G, R = n_groups, rows_per_group
    N = G * R
    xs = [torch.randn(N, dtype=torch.float64, device=device, generator=gen) for _ in range(p)]
    y = torch.zeros(N, dtype=torch.float64, device=device)
    for j in range(p):
        bj = torch.randn(G, dtype=torch.float64, device=device, generator=gen)
        y.add_(xs[j] * bj.repeat_interleave(R))
        del bj
    y.add_(torch.randn(N, dtype=torch.float64, device=device, generator=gen), alpha=0.1)
    return xs, y
