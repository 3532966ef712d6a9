"""numpy model of the FAST-mode E-step algorithm (tiled sweeps: speculate,
verify at the tile boundaries, repair only where the chain had not forgotten
its start; lagged normalisation; counts as a GEMM over bins).

TEST INFRASTRUCTURE: an executable specification of what the fast HIP kernels
(psmc_amd/csrc/estep_fast.hip) compute, used on CPU to check the algebra
against the oracle and to study tolerance / work as a function of the tile
length T, the speculative warm-up W0 and the tolerances.  Not imported by the
product.

Notation (1-indexed positions p=1..L, o_p the observation):
  forward  X_p = e[o_p] * (a^T X_{p-1}) / d_p ,  d_p = sum(X_{p-1}) if p % NORM_EVERY == 0 else 1
           (LL = sum_p log d_p + log sum(X_L) telescopes for any positive d_p)
  backward bt_p = e[o_p] * (a bt_{p+1}) * sb_p,  sb_p = 1/sum(bt_{p+1}) if p % NORM_EVERY == 0 else 1
           (independent of the forward tables: the two sweeps run concurrently)
  counts   G_p = sum_k X_p bt_p / e[o_p]                    (posterior normaliser, per position)
           C  += (sb_p / G_p) X_p (x) bt_{p+1}             p = 1..L-1   (A = a .* C)
           E[o_p] += X_p * bt_p / e[o_p] / G_p               p = 1..L-1
Speculation: a tile [lo,hi] starts W0 bins outside itself from an arbitrary
vector.  Verification compares the vector a tile used at its boundary with the
value its neighbour computed with a whole tile of history behind it; tiles
whose mismatch exceeds `tol` are re-run from the neighbour's value until the new
trajectory meets the stored one to `tol` again (or the tile ends, which may make
the next tile dirty).  Both sweeps are self-normalising, so stored vectors of
neighbouring tiles agree in scale as well as in direction.
"""
import numpy as np

TINY = 1e-25
NORM_EVERY = 4  # psmc_hip_internal.h


def plan_chunks(L, T):
    """[(lo, hi)] tiles of 1..L of length T (last one shorter)."""
    out = []
    lo = 1
    while lo <= L:
        hi = min(L, lo + T - 1)
        out.append((lo, hi))
        lo = hi + 1
    return out


def _relmax(x, y):
    return np.abs(x - y).max() / np.abs(y).max()


def estep_fast_model(a, e, a0, segs, T=4096, W=2048, tol=None, stats=None):
    """tol=None: pure warm-up tiling (no verification).  tol=float: speculate with W, verify, repair."""
    n = a.shape[0]
    A = np.zeros((n, n)); E = np.zeros((3, n)); LL = 0.0
    work = dict(fwd_steps=0, bwd_steps=0, fwd_rounds=0, bwd_rounds=0, bins=0)
    for seg in segs:
        seg = np.asarray(seg, dtype=np.int64)
        L = len(seg)
        work["bins"] += L
        o = np.concatenate([[2], seg, [2]])  # o[p], p=1..L
        tiles = plan_chunks(L, T)
        nc = len(tiles)
        X = np.zeros((L + 2, n)); d = np.ones(L + 2); bt = np.zeros((L + 2, n))

        def fstep(x, p):
            dp = x.sum() if p % NORM_EVERY == 0 else 1.0
            g = (x * e[o[p]]) if p == 1 else e[o[p]] * (x @ a)
            return g / dp, dp

        # ---------------- forward: speculative pass
        entry = [None] * nc
        for c, (lo, hi) in enumerate(tiles):
            ws = max(1, lo - W)
            x = a0.copy()
            for p in range(ws, hi + 1):
                if p == lo:
                    entry[c] = x.copy()
                x, dp = fstep(x, p)
                work["fwd_steps"] += 1
                if p >= lo:
                    X[p] = x; d[p] = dp
        # ---------------- forward: verify / repair rounds
        if tol is not None:
            while True:
                dirty = [c for c in range(1, nc) if _relmax(entry[c], X[tiles[c][0] - 1]) > tol]
                if not dirty:
                    break
                work["fwd_rounds"] += 1
                for c in dirty:
                    lo, hi = tiles[c]
                    x = X[lo - 1].copy(); entry[c] = x.copy()
                    for p in range(lo, hi + 1):
                        xn, dp = fstep(x, p)
                        work["fwd_steps"] += 1
                        done = _relmax(xn, X[p]) <= tol
                        X[p] = xn; d[p] = dp; x = xn
                        if done:
                            break
        LL += np.log(d[2:L + 1]).sum() + np.log(X[L].sum())

        # ---------------- backward: independent of the forward tables (own lagged scaling)
        sb = np.ones(L + 2)

        def bstep(btn, p):  # consumes bt_{p+1}; returns bt_p and the scale it applied
            sp = 1.0 / btn.sum() if p % NORM_EVERY == 0 else 1.0
            return e[o[p]] * (a @ btn) * sp, sp

        bentry = [None] * nc; bexit = [None] * nc
        for c, (lo, hi) in enumerate(tiles):
            top = min(hi, L - 1)
            if top < lo:
                continue
            q = min(hi + W + 1, L)
            btn = e[o[q]] * np.ones(n)
            for p in range(q - 1, lo - 1, -1):
                if p == top:
                    bt[top + 1] = btn; bentry[c] = btn.copy()
                btn, sp = bstep(btn, p)
                work["bwd_steps"] += 1
                if p <= top:
                    sb[p] = sp
                    if p > lo or p == 1:
                        bt[p] = btn
                    if p == lo:
                        bexit[c] = btn.copy()
        if tol is not None:
            while True:
                dirty = [c for c in range(nc - 1)
                         if bexit[c + 1] is not None and bentry[c] is not None
                         and tiles[c][1] + W + 1 < L and _relmax(bentry[c], bexit[c + 1]) > tol]
                if not dirty:
                    break
                work["bwd_rounds"] += 1
                for c in dirty:
                    lo, hi = tiles[c]
                    top = min(hi, L - 1)
                    btn = bexit[c + 1].copy(); bentry[c] = btn.copy()
                    bt[top + 1] = btn
                    for p in range(top, lo - 1, -1):
                        btn, sp = bstep(btn, p)
                        work["bwd_steps"] += 1
                        sb[p] = sp
                        if p > lo:
                            done = _relmax(btn, bt[p]) <= tol
                            bt[p] = btn
                            if done:
                                break
                        else:
                            bexit[c] = btn.copy()
                            if p == 1:
                                bt[1] = btn
        # ---------------- counts (GEMM over bins) from the stored tables, normalised per position:
        #   G_p = sum_k X_p bt_p / e[o_p]   (= sb_p * sum_kl X_p[k] a[k][l] bt_{p+1}[l])
        #   gamma_p = X_p bt_p / e[o_p] / G_p ;  xi_p = (sb_p / G_p) X_p (x) bt_{p+1} .* a
        if L > 1:
            re = np.where(e > 0, 1.0 / np.where(e > 0, e, 1.0), 0.0)
            g = X[1:L] * bt[1:L] * re[seg[:L - 1]]
            G = g.sum(1)
            C = (X[1:L] * (sb[1:L] / G)[:, None]).T @ bt[2:L + 1]
            A += a * C
            S = np.zeros((3, n))
            gam = g / G[:, None]
            for b in range(3):
                S[b] = gam[seg[:L - 1] == b].sum(0)
            E += S
        A += TINY; E += TINY
    if stats is not None:
        stats.update(work)
    return dict(A=A, E=E[:2].copy(), LL=LL)


# ---------------------------------------------------------------------------------------------
# Structured O(N) sweeps (psmc_amd/csrc/estep_struct.hip, api.hip factor_structure): numpy mirror.
def factor_structure(a, ulps=64):
    """a[k][l] = P_k qa_l (l<k), R_k c_l (l>k) with qa_0 = c_{n-1} = 1, dd = diag - P.qa - R.c >= 0,
    every off-diagonal entry checked to `ulps`; returns None when the matrix does not have the form
    (lh3/psmc core.c:112-122 builds it this way; psmc_cap_matrix, aux.c:115-127, destroys it)."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    if n < 3 or not (a[n - 1, 0] > 1e-280 and a[0, n - 1] > 1e-280):
        return None
    P = np.zeros(n); R = np.zeros(n); qa = np.zeros(n); c = np.zeros(n)
    qa[:n - 1] = a[n - 1, :n - 1] / a[n - 1, 0]
    P[1:] = a[1:, 0]
    c[1:] = a[0, 1:] / a[0, n - 1]
    R[:n - 1] = a[:n - 1, n - 1]
    w = np.where(np.tri(n, k=-1, dtype=bool), np.outer(P, qa), np.outer(R, c))
    off = ~np.eye(n, dtype=bool)
    if not np.all(np.abs(a - w)[off] <= ulps * 2.220446049250313e-16 * np.abs(a)[off] + 1e-290):
        return None
    dd = np.diag(a) - P * qa - R * c
    if not np.all(dd >= 0):
        return None
    return dict(P=P, R=R, qa=qa, c=c, dd=dd)


def struct_step_forward(f, x):
    """(a^T x)_j = qa_j SUF_j(x.P) + c_j PRE_j(x.R) + dd_j x_j, inclusive suffix / prefix sums."""
    return f["qa"] * np.cumsum((x * f["P"])[::-1])[::-1] + f["c"] * np.cumsum(x * f["R"]) + f["dd"] * x


def struct_step_backward(f, z):
    """(a z)_k = R_k SUF_k(z.c) + P_k PRE_k(z.qa) + dd_k z_k."""
    return f["R"] * np.cumsum((z * f["c"])[::-1])[::-1] + f["P"] * np.cumsum(z * f["qa"]) + f["dd"] * z
