"""Development aid: randomised shapes through the grouped paths against the oracle (fused / centred bias / packed / ridge)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 20
n_cfg = worst = 0
while time.time() < t_end:
    p = int(rng.integers(1, 17)); bias = bool(rng.integers(0, 2)); lam = float(rng.choice([0.0, 0.0, 0.3]))
    G = int(rng.integers(1, 400)); hi = int(rng.choice([3, 40, 300, 2000]))
    sizes = rng.integers(0, hi, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    if N == 0: continue
    X = rng.normal(size=(N, p)) + rng.normal(size=p) * float(rng.choice([0.0, 3.0]))
    y = X @ rng.normal(size=p) + rng.normal(size=N) * 0.1 + 0.5
    dt = np.float64
    cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]
    co, nu = pds.lin_reg_by(*cols, target=torch.from_numpy(y).cuda(), group_offsets=off, add_bias=bias, l2_reg=lam)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=lam, nthreads=8)
    pp = p + bias
    well = (~nu) & (~nu_o) & (sizes >= 3 * pp + 10)
    edge = (nu != nu_o) & (sizes >= 3 * pp + 10)   # gate decisions may only differ on groups near the threshold
    assert not edge.any(), (p, bias, lam, np.flatnonzero(edge)[:5], sizes[edge][:5])
    assert np.array_equal(nu[sizes < pp], np.ones((sizes < pp).sum(), bool))
    if well.any():
        err = np.linalg.norm(co[well] - co_o[well], axis=1) / np.linalg.norm(co_o[well], axis=1)
        worst = max(worst, float(err.max()))
        assert err.max() < 1e-8, (p, bias, lam, float(err.max()))
    n_cfg += 1
print(f"{n_cfg} random configurations ok, worst normwise rel. error on well-determined groups {worst:.2e}")
