#!/usr/bin/env python3
"""Exact mode at ANY number of hidden states against the CPU oracle, bit for bit (round 6; `python scripts/fuzz_gpu_anyn.py SECONDS
[SEED0]`).  Every case: n uniform in 1..420 (the 64-state, 128-state, register-resident 129..224 and general wide kernels all come
up; every size that is not a multiple of 64 is padded), a random dense HMM (no PSMC structure), 1-6 segments of 1..1500 positions,
a multiset selection; checked: A, E, A0, LL, the per-segment check sums, decoding and posterior of one segment, a batch of three
replicates."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psmc_amd import hip
import orc


def bits_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    return x.shape == y.shape and bool((x.view(np.uint64) == y.view(np.uint64)).all())


def random_hmm(rng, n):
    a = rng.random((n, n)) ** 4 * 0.02 + np.eye(n) * (0.9 + 0.1 * rng.random(n))
    a /= a.sum(1, keepdims=True)
    e = np.ones((3, n)); e[1] = 0.001 + rng.random(n) * 0.15; e[0] = 1.0 - e[1]
    a0 = rng.random(n) + 0.1; a0 /= a0.sum()
    return a, e, a0


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = orc.Oracle()
    t_end = time.time() + budget
    stats = dict(cases=0, failures=[], n_seen=[])
    seed = seed0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 421))
        segs = [rng.choice(3, size=int(np.exp(rng.uniform(0, np.log(1500)))), p=[0.86, 0.1, 0.04]).astype(np.uint8) for _ in range(int(rng.integers(1, 7)))]
        sel = rng.integers(0, len(segs), size=int(rng.integers(1, len(segs) + 3))).tolist()
        a, e, a0 = random_hmm(rng, n)
        case = dict(seed=seed, n=n, segs=[len(s) for s in segs], sel=sel)
        try:
            es = hip.HipEStep(n, mode=hip.MODE_EXACT)
            es.load_segments(segs); es.select(sel)
            r = es.estep(a, e, a0)
            o = oracle.estep(a, e, a0, [segs[i] for i in sel], per_seg=True)
            what = []
            if not (bits_equal(r["A"], o["A"]) and bits_equal(r["E"], o["E"]) and bits_equal(r["A0"], o["A0"]) and r["LL"] == o["LL"]): what.append("statistics")
            if not bits_equal(r["chk"], o["seg_chk"]): what.append("check sums")
            k = sel[int(rng.integers(len(sel)))]
            f, b, s, lk, chk = oracle.fwd_bwd(a, e, a0, segs[k])
            path, mp = oracle.post_decode(f, b, s); gp, gm = es.decode(k)
            post, rec = oracle.post_full(a, e, segs[k], f, b, s); pp, rr = es.posterior(k)
            if not (np.array_equal(gp, path[1:]) and bits_equal(gm, mp[1:])): what.append("decode")
            if not (bits_equal(pp, post[1:]) and bits_equal(rr, rec[1:])): what.append("posterior")
            pars = [random_hmm(rng, n) for _ in range(3)]
            sels = [rng.integers(0, len(segs), size=int(rng.integers(1, len(segs) + 2))).tolist() for _ in range(3)]
            got = es.estep_batch(pars, sels)
            for q in range(3):
                w = oracle.estep(pars[q][0], pars[q][1], pars[q][2], [segs[i] for i in sels[q]])
                if not (bits_equal(got["A"][q], w["A"]) and bits_equal(got["E"][q], w["E"]) and got["LL"][q] == w["LL"]): what.append("batch replicate %d" % q)
            es.close()
            if what:
                stats["failures"].append(dict(case, what=what)); print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
        except Exception as err:
            stats["failures"].append(dict(case, error=repr(err)[:300])); print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
        stats["cases"] += 1; stats["n_seen"].append(n)
        seed += 1
    ns = np.array(stats.pop("n_seen"))
    stats["n_by_kernel"] = {"1..64": int((ns <= 64).sum()), "65..128": int(((ns > 64) & (ns <= 128)).sum()), "129..224": int(((ns > 128) & (ns <= 224)).sum()), "225..420": int((ns > 224).sum())}
    stats["seeds"] = [seed0, seed - 1]
    print(json.dumps(stats, indent=1))


if __name__ == "__main__":
    main()
