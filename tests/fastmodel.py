"""numpy model of the FAST-mode E-step algorithm (chunked sweeps with warm-up
overlap, lagged normalisation, per-chunk posterior normalisation).

TEST INFRASTRUCTURE: an executable specification of what the fast HIP kernels
(psmc_amd/csrc/estep_fast.hip) compute, used on CPU to (1) check the algebra
against the oracle and (2) quantify the tolerance as a function of the chunk
length T and warm-up W.  Not imported by the product.

Notation (1-indexed positions p=1..L, o_p the observation):
  forward  X_p = e[o_p] * (a^T X_{p-1}) / d_p ,  d_p = sum(X_{p-1}),  d_1 = 1
           (so sum(X_p) = true scale s_p; LL = sum_p log(sum X_p))
  backward Bnew_p = a (e[o_{p+1}] * B_{p+1}),  B_p = Bnew_p / d_p
  counts   C += X_p (x) (e[o_{p+1}] * B_{p+1})   for p = 1..L-1
           E[o_p] += X_p * Bnew_p                for p = 1..L-1
  A = a * C   (elementwise), plus n_seg * HMM_TINY on every cell.
Each backward chunk is normalised once at its top position so that
sum_k X_top[k]*Bnew_top[k] = 1 (posterior sums to one); after that all products
inside the chunk are correctly scaled because forward and backward share d_p.
"""
import numpy as np

TINY = 1e-25


def plan_chunks(L, T):
    """[(lo, hi)] tiles of 1..L of length T (last one shorter)."""
    out = []
    lo = 1
    while lo <= L:
        hi = min(L, lo + T - 1)
        out.append((lo, hi))
        lo = hi + 1
    return out


def estep_fast_model(a, e, a0, segs, T=4096, W=2048, return_chunks=False):
    n = a.shape[0]
    A = np.zeros((n, n)); E = np.zeros((3, n)); LL = 0.0
    max_warm_err = 0.0
    for seg in segs:
        seg = np.asarray(seg, dtype=np.int64)
        L = len(seg)
        o = np.concatenate([[2], seg])  # o[p], p=1..L
        chunks = plan_chunks(L, T)
        nc = len(chunks)
        lo = np.array([c[0] for c in chunks]); hi = np.array([c[1] for c in chunks])
        X_store = np.zeros((L + 2, n)); d_store = np.ones(L + 2)
        # ---------------- forward: all chunks in lockstep
        ws = np.maximum(1, lo - W)             # first position computed by the chunk
        X = np.tile(a0, (nc, 1))               # "X_{ws-1}" prior (true a0 when ws==1)
        steps = (hi - ws + 1).max()
        ll_chunk = np.zeros(nc)
        for t in range(steps):
            p = ws + t
            act = p <= hi
            pc = np.minimum(p, L)
            em = e[o[pc]]                      # (nc, n)
            first = (p == 1)
            d = np.where(first, 1.0, X.sum(1))
            G = np.where(first[:, None], X * em, em * (X @ a))
            Xn = G / d[:, None]
            X = np.where(act[:, None], Xn, X)
            st = act & (p >= lo)
            X_store[pc[st]] = Xn[st]
            d_store[pc[st]] = d[st]
            ll_chunk[st] += np.log(Xn[st].sum(1))
        LL += ll_chunk.sum()
        # warm-up quality: compare each chunk's entry vector with the truth is
        # not available here; instead report |X_store| continuity via oracle in tests
        # ---------------- backward: all chunks in lockstep
        top = np.minimum(hi, L - 1)            # first accumulating position
        q = np.minimum(hi + W + 1, L)          # position whose B is initialised to 1
        B = np.ones((nc, n))
        C = np.zeros((nc, n, n)); Ec = np.zeros((nc, 3, n))
        steps = (q - lo).max() if nc else 0
        for t in range(steps):
            p = q - 1 - t                      # position being produced
            act = (p >= lo) & (p >= 1)
            pc = np.clip(p, 1, L)
            Bt = e[o[np.minimum(pc + 1, L)]] * B
            Bnew = Bt @ a.T
            Xp = X_store[pc]
            istop = act & (p == top)
            if istop.any():
                c = (Xp * Bnew).sum(1)
                kappa = np.where(istop, 1.0 / c, 1.0)
                Bt = Bt * kappa[:, None]; Bnew = Bnew * kappa[:, None]
            acc = act & (p <= top)
            if acc.any():
                C[acc] += Xp[acc][:, :, None] * Bt[acc][:, None, :]
                sym = o[pc]
                for b in range(3):
                    m = acc & (sym == b)
                    Ec[m, b] += Xp[m] * Bnew[m]
            Bn = Bnew / d_store[pc][:, None]
            B = np.where(act[:, None], Bn, B)
        A += a * C.sum(0) + TINY
        E += Ec.sum(0) + TINY
    out = dict(A=A, E=E[:2].copy(), LL=LL)
    return out
