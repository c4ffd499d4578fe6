"""Lane-level model of rolling_pair_kernel (csrc/rolling_pair_dev.hpp): checks the slot tables, the parity
permutation, the rotate-and-scan over the 8 chains of a 16-lane row and the P / Q alternation against a direct
computation.  Pure numpy, no GPU: `python tools/models/rolling_pair_model.py`.

Design under test (one 16-lane DPP row = one sub-stream; lanes (2c, 2c+1) = chain c; chain = K = 4 consecutive rows):
  * even lanes label the variables naturally, odd lanes reversed (pi(a) = PP-1-a); a lane keeps the running moments of
    the label pairs in the set A only, so the two lanes of a chain hold complementary halves (the PP/2 pairs {a, pi(a)}
    are held by both);
  * the work copy of a lane's own row is its own half + the partner's half (DPP quad_perm [1,0,3,2]);
  * even lanes solve the even rows of the chain, odd lanes the odd rows, in the same instruction stream.
"""
import numpy as np

PP = 8
K = 4


def tables(PP):
    pairs = [(a, b) for a in range(PP) for b in range(a, PP)]
    lin = {e: i for i, e in enumerate(pairs)}
    pi = lambda e: tuple(sorted((PP - 1 - e[0], PP - 1 - e[1])))
    A = []
    for e in pairs:
        f = pi(e)
        if f == e or e < f:
            A.append(e)
    fixed = [e for e in A if pi(e) == e]
    assert len(A) == PP * PP // 4 + PP // 2 and len(fixed) == PP // 2
    # gather: label entry e not in A comes from the partner's slot k with A[k] == pi(e)
    gather = {}
    for e in pairs:
        if e not in A:
            gather[e] = A.index(pi(e))
    return pairs, lin, pi, A, gather


def direct(X, y, w, lam):
    n, p = X.shape
    out = np.full((n, p), np.nan)
    for r in range(w - 1, n):
        Z = X[r - w + 1:r + 1]
        G = Z.T @ Z + lam * np.eye(p)
        out[r] = np.linalg.solve(G, Z.T @ y[r - w + 1:r + 1])
    return out


def model(X, y, w, lam, t0, nrows):
    """One sub-stream (16 lanes) walks rows [t0, t0 + nrows) in stages of 32 rows; returns beta rows."""
    pairs, lin, pi, A, gather = tables(PP)
    NA = len(A)
    H = PP // 2
    NS = NA + H + 1
    n = X.shape[0]
    lanes = 16
    par = np.arange(lanes) & 1
    ch = np.arange(lanes) >> 1

    def row_regs(r):  # per lane: label-ordered row values (odd lanes reversed); zeros outside the frame
        z = np.zeros((lanes, PP))
        yy = np.zeros(lanes)
        for l in range(lanes):
            rr = r[l]
            if 0 <= rr < n:
                z[l] = X[rr] if par[l] == 0 else X[rr][::-1]
                yy[l] = y[rr]
        return z, yy

    def inc(dst, src, zn, yn, zo, yo, okn, oko):
        for k, (a, b) in enumerate(A):
            dst[:, k] = src[:, k] + zn[:, a] * zn[:, b] - zo[:, a] * zo[:, b]
        for a in range(H):
            dst[:, NA + a] = src[:, NA + a] + zn[:, a] * yn - zo[:, a] * yo
        dst[:, NS - 1] = src[:, NS - 1] + okn - oko

    # anchor: every lane holds its half of the window in front of the tile
    P = np.zeros((lanes, NS))
    for r in range(t0 - w, t0):
        zn, yn = row_regs(np.full(lanes, r))
        ok = float(0 <= r < n)
        inc(P, P.copy(), zn, yn, np.zeros((lanes, PP)), np.zeros(lanes), ok, 0.0)
    Q = np.zeros((lanes, NS))
    out = {}
    for base in range(t0, t0 + nrows, 32):
        r0 = base + 4 * ch
        # pass 1
        Q[:] = 0.0
        for i in range(K):
            zn, yn = row_regs(r0 + i)
            zo, yo = row_regs(r0 + i - w)
            okn = ((r0 + i) < n).astype(float)
            oko = ((r0 + i - w) >= 0).astype(float) * okn
            zo *= oko[:, None]; yo = yo * oko
            inc(Q, Q.copy(), zn, yn, zo, yo, okn, oko)
        # select + rotate + scan (stride-2 shifts keep the parity)
        U = np.where((ch == 7)[:, None], P, Q)
        Y = np.roll(U, 2, axis=0)          # row_ror:2 -- lane i <- lane (i - 2) mod 16
        for s in (2, 4, 8):
            sh = np.zeros_like(Y); sh[s:] = Y[:-s]   # row_shr:s, zero fill
            Y = Y + sh
        P = Y
        # pass 2
        for m in range(K // 2):
            zn0, yn0 = row_regs(r0 + 2 * m)
            zo0, yo0 = row_regs(r0 + 2 * m - w)
            ok0 = ((r0 + 2 * m) < n).astype(float); oo0 = ((r0 + 2 * m - w) >= 0).astype(float) * ok0
            zo0 *= oo0[:, None]; yo0 = yo0 * oo0
            inc(Q, P, zn0, yn0, zo0, yo0, ok0, oo0)          # step A: Q = P + inc(row 2m)
            zn1, yn1 = row_regs(r0 + 2 * m + 1)
            zo1, yo1 = row_regs(r0 + 2 * m + 1 - w)
            ok1 = ((r0 + 2 * m + 1) < n).astype(float); oo1 = ((r0 + 2 * m + 1 - w) >= 0).astype(float) * ok1
            zo1 *= oo1[:, None]; yo1 = yo1 * oo1
            Pn = np.zeros_like(P)
            inc(Pn, Q, zn1, yn1, zo1, yo1, ok1, oo1)         # step B: P = Q + inc(row 2m+1)
            P = Pn
            own = np.where(par[:, None] == 1, P, Q)          # even lanes: state after row 2m; odd lanes: after 2m+1
            send = np.where(par[:, None] == 1, Q, P)         # what the PARTNER needs: its state, my half
            partner = send.reshape(8, 2, NS)[:, ::-1, :].reshape(lanes, NS)   # quad_perm [1,0,3,2]
            G = np.zeros((lanes, PP, PP)); c = np.zeros((lanes, PP))
            for e in pairs:
                v = own[:, A.index(e)] if e in A else partner[:, gather[e]]
                G[:, e[0], e[1]] = v; G[:, e[1], e[0]] = v
            for a in range(PP):
                c[:, a] = own[:, NA + a] if a < H else partner[:, NA + (PP - 1 - a)]
            for l in range(lanes):
                r = r0[l] + 2 * m + par[l]
                if r >= n:
                    continue
                beta = np.linalg.solve(G[l] + lam * np.eye(PP), c[l])
                out[r] = beta if par[l] == 0 else beta[::-1]     # odd lanes: back to the true order
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n, w, lam = 700, 50, 0.1
    X = rng.normal(size=(n, PP)); y = X @ rng.normal(size=PP) + 0.1 * rng.normal(size=n)
    ref = direct(X, y, w, lam)
    worst = 0.0
    for t0, nr in ((0, 256), (256, 256), (512, 192)):
        got = model(X, y, w, lam, t0, nr)
        for r, b in got.items():
            if r >= w - 1:
                worst = max(worst, np.max(np.abs(b - ref[r]) / (1e-12 + np.abs(ref[r]))))
    print("max rel diff vs direct window solves:", worst)
    assert worst < 1e-9
    pairs, lin, pi, A, gather = tables(PP)
    print("A =", A)
    print("gather =", gather)
