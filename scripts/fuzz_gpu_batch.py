#!/usr/bin/env python3
"""Randomised run of the exact batch (psmc_hip_estep_batch_cb) against separate exact E-steps, bit for bit (round 6; companion of
scripts/fuzz_gpu.py: `python scripts/fuzz_gpu_batch.py SECONDS [SEED0]`).  Every case: 3-40 trunks (most of one length, like
utils/splitfa.c's, some longer tails, some short), 2-24 replicates drawn with replacement, a random schedule -- table memory for
1/1 .. 1/6 of the bins per launch, a share of the compute units (entry slots), with and without the f table, "batch_sort" /
"batch_tailfill" / "batch_major" on or off, with and without the progress callback, sometimes a reservation in two calls
(second table chunk).  Checked: A, E, LL of every replicate equal the bits of select() + estep(); `done` names every replicate once."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip, hostlib


def bits_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    return x.shape == y.shape and bool((x.view(np.uint64) == y.view(np.uint64)).all())


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
    P = [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"][1:]]
    t_end = time.time() + budget
    stats = dict(cases=0, replicates=0, launches=0, two_chunks=0, failures=[])
    seed = seed0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        base = int(rng.choice([500, 2000, 5000]))
        Ls = []
        for _ in range(int(rng.integers(3, 41))):
            u = rng.random()
            Ls.append(base if u < 0.6 else (int(rng.integers(base, 3 * base // 2)) if u < 0.85 else int(rng.integers(1, base))))
        trunks = [rng.choice(np.array([0, 0, 0, 0, 1, 2], dtype=np.uint8), size=l) for l in Ls]
        n_rep = int(rng.integers(2, 25))
        sels = [rng.integers(0, len(Ls), size=int(rng.integers(1, len(Ls) + 1))).tolist() for _ in range(n_rep)]
        params = [P[int(rng.integers(len(P)))] for _ in range(n_rep)]
        bins = sum(sum((Ls[i] + 63) // 64 * 64 for i in set(x)) for x in sels)
        entries = sum(len(set(x)) for x in sels)
        opts = dict(exact_refwd=int(rng.choice([0, 1, 2])))
        frac = int(rng.integers(1, 7))
        two = False
        if rng.random() < 0.75: opts["batch_bins"] = max(bins // frac + 4096, 4 * (max(Ls) + 64) + 4096)   # (a block of four entries must fit)
        elif opts["exact_refwd"] >= 1 and rng.random() < 0.6: two = True
        for k in ("batch_sort", "batch_tailfill", "batch_major"):
            if rng.random() < 0.3: opts[k] = 0
        cus = int(rng.choice([0, 8, max(2, (entries // frac + 8) // 4), 64]))
        use_cb = rng.random() < 0.6
        case = dict(seed=seed, trunks=len(Ls), base=base, n_rep=n_rep, entries=entries, bins=bins, opts=opts, cus=cus, cb=use_cb, two=two)
        try:
            es = hip.HipEStep(64, mode=hip.MODE_EXACT, **opts)
            if cus: es.set_cu_range(0, min(cus, 256))
            es.load_segments(trunks)
            if two:
                es.reserve_batch_tables(max(bins // 3, 2 * max(Ls) + 4096)); es.reserve_batch_tables(bins); stats["two_chunks"] += 1
            seen = []
            got = es.estep_batch(params, sels, on_done=(lambda reps, out: seen.extend(reps)) if use_cb else None)
            stats["launches"] += es.batch_info()["groups"]
            es.close()
            if use_cb and sorted(seen) != list(range(n_rep)):
                raise AssertionError("done named %s" % sorted(seen))
            ref = hip.HipEStep(64, mode=hip.MODE_EXACT); ref.load_segments(trunks)
            for r in rng.choice(n_rep, size=min(n_rep, 4), replace=False):
                ref.select(sels[r]); w = ref.estep(*params[r])
                if not (bits_equal(got["A"][r], w["A"]) and bits_equal(got["E"][r], w["E"]) and got["LL"][r] == w["LL"]):
                    raise AssertionError("replicate %d differs from its separate E-step" % r)
            ref.close()
        except Exception as err:
            stats["failures"].append(dict(case, error=repr(err)[:300]))
            print("FAIL", json.dumps(stats["failures"][-1], default=int), flush=True)
        stats["cases"] += 1; stats["replicates"] += n_rep
        seed += 1
    stats["seeds"] = [seed0, seed - 1]
    print(json.dumps(stats, indent=1, default=int))


if __name__ == "__main__":
    main()
