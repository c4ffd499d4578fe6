"""
NumPy model of tools/experiments/rolling_seg.hip's index logic (tiles, anchor, stages, lane segments, masks): the window
moments every (tile, stage, lane, row) ends up with must equal the direct window sums.  CPU only; run it after touching the
kernel's bookkeeping.   python tools/experiments/rolling_seg_model.py
"""
import numpy as np

TILE, LANES = 4096, 64


def moments(z, y, ok):
    p = z.shape[1]
    iu = np.triu_indices(p)
    m = np.concatenate([(z[:, :, None] * z[:, None, :])[:, iu[0], iu[1]], z * y[:, None], ok[:, None].astype(float)], axis=1)
    return np.where(ok[:, None], m, 0.0)


def model(X, y, w, K):
    n, p = X.shape
    ok = np.isfinite(X).all(axis=1) & np.isfinite(y)
    M = moments(np.where(ok[:, None], X, 0.0), np.where(ok, y, 0.0), ok)          # m(r)
    nv = M.shape[1]

    def m_at(r, limit):  # load_rows: rows outside [0, limit) are zero
        return M[r] if 0 <= r < limit else np.zeros(nv)

    out = np.full((n, nv), np.nan)
    for t0 in range(0, n, TILE):
        t1 = min(t0 + TILE, n)
        carry = np.zeros(nv)
        for lane in range(LANES):                                                  # anchor: rows t0 - w ... t0 - 1 in pairs
            r = t0 - w + 2 * lane
            while r < t0:
                limit = n if r + 1 < t0 else min(r + 1, n)
                carry = carry + m_at(r, limit) + m_at(r + 1, limit)
                r += 128
        for base in range(t0, t1, LANES * K):
            D = np.zeros((LANES, nv))
            for lane in range(LANES):
                for i in range(K):
                    r = base + K * lane + i
                    new = m_at(r, t1)
                    old = m_at(r - w, n) if r < t1 else np.zeros(nv)
                    D[lane] += new - old
            incl = np.cumsum(D, axis=0)
            start = carry + incl - D
            carry = carry + incl[-1]
            for lane in range(LANES):
                S = start[lane].copy()
                for i in range(K):
                    r = base + K * lane + i
                    new = m_at(r, t1)
                    old = m_at(r - w, n) if r < t1 else np.zeros(nv)
                    S = S + new - old
                    if r < t1:
                        out[r] = S
    return out, M


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, p, w, K in [(9000, 3, 256, 4), (4097, 2, 255, 4), (5000, 2, 7, 2), (300, 3, 64, 4), (8192, 1, 4096, 4), (100, 2, 200, 4)]:
        X = rng.normal(size=(n, p))
        y = rng.normal(size=n)
        X[rng.random(n) < 0.01, 0] = np.nan
        got, M = model(X, y, w, K)
        c = np.concatenate([np.zeros((1, M.shape[1])), np.cumsum(M, axis=0)])
        lo = np.maximum(np.arange(n) - w + 1, 0)
        want = c[np.arange(n) + 1] - c[lo]
        err = np.abs(got - want).max()
        print(f"n={n} p={p} w={w} K={K}: max |window moments - direct| = {err:.2e}")
        assert err < 1e-8
    print("index logic ok")
