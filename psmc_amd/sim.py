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


# hg19 autosome lengths in Mb (public assembly statistics): the shape of a whole-genome .psmcfa
_AUTOSOMES_MB = [249.25, 243.20, 198.02, 191.15, 180.92, 171.12, 159.14, 146.36, 141.21, 135.53, 135.01, 133.85,
                 115.17, 107.35, 102.53, 90.35, 81.20, 78.08, 59.13, 63.03, 48.13, 51.30]


def human_like_lengths(total_bins, n_seg=90, longest_frac=0.083):
    """n_seg segment lengths summing to ~total_bins, shaped like a human genome in 100-bp bins: the 22 autosomes in
    hg19 proportions with the longest at longest_frac of the total (3e7 bins -> chr1 = 2.49e6, SURVEY.md section 8(d)
    config 3), the remainder spread over small scaffolds of linearly decaying size."""
    chrom = np.array(_AUTOSOMES_MB[:min(n_seg, len(_AUTOSOMES_MB))])
    chrom = chrom / chrom[0] * (longest_frac * total_bins)
    n_small = n_seg - len(chrom)
    rest = total_bins - chrom.sum()
    if n_small <= 0 or rest <= n_small:     # no room (or no wish) for scaffolds: the chromosomes alone, rescaled
        chrom = chrom * (total_bins / chrom.sum())
        small = np.zeros(0)
    else:
        w = np.linspace(1.0, 0.1, n_small)
        small = w / w.sum() * rest
    L = np.maximum(1, np.round(np.concatenate([chrom, small])).astype(np.int64))
    return L.astype(np.int32)


def simulate_genome(a, e, a0, lengths, seed):
    rng = np.random.default_rng(seed)
    return [simulate_segment(a, e, a0, int(L), rng) for L in lengths]
