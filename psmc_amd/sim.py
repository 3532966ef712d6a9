"""Seeded synthetic .psmcfa-like observation streams drawn from a PSMC HMM.

Plays the role hmm_simulate (lh3/psmc khmm.c:386-423) plays for the reference's
`-S` option -- hidden state path from (a0, a), het/hom emission from e -- but as
a jump chain: the transition matrix is strongly diagonal, so run lengths are
drawn geometrically and only the ~2 % of bins where the state changes cost a
Python iteration.  Used for benchmark / test inputs only (no data files can be
fetched); it is not part of the E-step path.
"""
import numpy as np


def simulate_segment(a, e, a0, L, rng, miss_rate=0.0004, miss_lo=10, miss_hi=90):
    """Return L bytes in {0 hom, 1 het, 2 missing} (psmc_seq_t.seq encoding, cli.c:15-32)."""
    n = a.shape[0]
    diag = np.clip(np.diag(a), 0.0, 1.0 - 1e-12)
    offd = a.copy()
    np.fill_diagonal(offd, 0.0)
    offc = np.cumsum(offd / offd.sum(1, keepdims=True), axis=1)
    states = np.empty(L, dtype=np.int16)
    k = int(min(np.searchsorted(np.cumsum(a0), rng.random()), n - 1))
    pos = 0
    while pos < L:
        run = int(rng.geometric(1.0 - diag[k]))
        end = min(L, pos + run)
        states[pos:end] = k
        pos = end
        k = int(min(np.searchsorted(offc[k], rng.random()), n - 1))
    seq = (rng.random(L) < e[1][states]).astype(np.uint8)
    if miss_rate > 0:
        starts = np.nonzero(rng.random(L) < miss_rate)[0]
        lens = rng.integers(miss_lo, miss_hi + 1, size=len(starts))
        for s, l in zip(starts, lens):
            seq[s:s + l] = 2
    return seq


def human_like_lengths(total_bins, n_seg=90, longest_frac=0.083):
    """n_seg segment lengths summing to ~total_bins, shaped like human chromosomes
    plus scaffolds: 24 large ones decaying from the longest, the rest small."""
    big = np.linspace(1.0, 0.19, 24)
    small = np.linspace(0.02, 0.002, max(n_seg - 24, 0))
    w = np.concatenate([big, small])[:n_seg]
    w = w / w.sum()
    L = np.maximum(1, np.round(w * total_bins).astype(np.int64))
    return L.astype(np.int32)


def simulate_genome(a, e, a0, lengths, seed):
    rng = np.random.default_rng(seed)
    return [simulate_segment(a, e, a0, int(L), rng) for L in lengths]
