#!/usr/bin/env python3
"""The two input files of the north-star job (README:49-62 of the reference), synthetic: genome30m.psmcfa -- 90 segments,
30 M bins, longest 2.49 M (bench.py's config 3) -- and split.psmcfa, what utils/splitfa.c makes of the 22-chromosome
version of the same genome (500 k-bin trunks, a tail shorter than 1.5 trunks stays whole).  Shared by scripts/northstar.py,
scripts/time_boot.py and the round-5 experiments, so that they all time the same bytes."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_psmcfa(path, segs, prefix):
    conv = np.frombuffer(b"TKN", dtype=np.uint8)
    with open(path, "wb") as fh:
        for i, s in enumerate(segs):
            fh.write((">%s%d\n" % (prefix, i)).encode())
            t = conv[s]
            n60 = len(t) // 60 * 60
            fh.write(np.concatenate([t[:n60].reshape(-1, 60), np.full((n60 // 60, 1), 10, np.uint8)], axis=1).tobytes())
            if n60 < len(t):
                fh.write(t[n60:].tobytes() + b"\n")


def params():
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
    return g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]


def genome(total=30_000_000):
    """(path-independent) the 90 segments of the main run"""
    from psmc_amd import sim
    a, e, a0 = params()
    return sim.simulate_genome(a, e, a0, sim.human_like_lengths(total, n_seg=90), seed=43)


def trunks(total=30_000_000):
    """splitfa of the 22-chromosome genome: utils/splitfa.c:20-35"""
    from psmc_amd import sim
    a, e, a0 = params()
    out = []
    for s in sim.simulate_genome(a, e, a0, sim.human_like_lengths(total, n_seg=22), seed=43):
        L, pos = len(s), 0
        while L - pos >= 750_000:
            out.append(s[pos:pos + 500_000]); pos += 500_000
        out.append(s[pos:])
    return out


def files(tmp=None, total=30_000_000, want=("genome", "split")):
    """write what is missing, return dict(genome=path, split=path, n_trunks, trunk_bins, longest_trunk)"""
    tmp = tmp or os.environ.get("TMPDIR", "/tmp")
    res = {}
    if "genome" in want:
        p = os.path.join(tmp, "genome30m.psmcfa" if total == 30_000_000 else "genome_%d.psmcfa" % total)
        if not os.path.exists(p):
            write_psmcfa(p, genome(total), "seg")
        res["genome"] = p
    if "split" in want:
        p = os.path.join(tmp, "split.psmcfa" if total == 30_000_000 else "split_%d.psmcfa" % total)
        meta = p + ".meta"
        if not os.path.exists(p) or not os.path.exists(meta):
            t = trunks(total)
            write_psmcfa(p, t, "t")
            open(meta, "w").write("%d %d %d\n" % (len(t), sum(len(x) for x in t), max(len(x) for x in t)))
        n, b, m = [int(x) for x in open(meta).read().split()]
        res.update(split=p, n_trunks=n, trunk_bins=b, longest_trunk=m)
    return res


if __name__ == "__main__":
    print(files())
