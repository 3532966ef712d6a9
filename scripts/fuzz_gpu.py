#!/usr/bin/env python3
"""Randomised differential run of the fast E-step against the exact one on the GPU (round 6).  Not part of the test-suite: a
campaign to run on a lease (`python scripts/fuzz_gpu.py SECONDS [SEED0]`), its summary kept under profiles/.

Every case: 1-12 segments with lengths log-uniform in [1, 300 k], simulated under a parameter set of the committed EM trajectory
(64 or 128 states), with planted runs of missing data (up to 60 k bins) and of homozygous bins; a random plan (chunk, warmup,
fused or not, gap tiles, the round-6 options merge / adapt / prev_start); four E-steps with DIFFERENT parameter sets on the same
context (the plan learns across them), full counts and factored sums, each compared with exact mode at the bounds the suite uses
(tests/test_gpu_estep.py check_fast).  PSMC_HIP_ECONVERGE is an allowed answer for plans that cannot converge (it is counted).
FUZZ_DEFAULT=1: every case with the default plan (what a caller without options gets).  FUZZ_BIG=1: segments of 20 k..1 M bins and
small tiles, so that the plans have more than 4096 tiles (two rounds of the fused back half)."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip, hostlib, sim
from psmc_amd.parity import fast_error_metrics

TOL = dict(A_max=1e-10, E_max=1e-10, LL=1e-12, A_cell=1e-9, E_cell=1e-9, A_l1=1e-10, QA=1e-10, QE=1e-10)


def traj(n):
    tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n%d.json" % n)))
    return [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"][1:]]


def make_segments(rng, p):
    segs = []
    for _ in range(int(rng.integers(1, 13))):
        L = int(np.exp(rng.uniform(np.log(20_000), np.log(1_000_000)))) if os.environ.get("FUZZ_BIG") else int(np.exp(rng.uniform(0, np.log(300_000))))
        s = sim.simulate_segment(p[0], p[1], p[2], L, rng)
        for _ in range(int(rng.integers(0, 4))):           # planted runs: missing data or homozygous
            if L < 8: break
            w = int(np.exp(rng.uniform(0, np.log(min(L, 60_000)))))
            at = int(rng.integers(0, L - w + 1))
            s[at:at + w] = 2 if rng.random() < 0.7 else 0
        segs.append(s)
    return segs


def sums_of(A):
    n = A.shape[0]
    lo, up = np.tril(A, -1), np.triu(A, 1)
    return np.stack([lo.sum(1), up.sum(1), np.diag(A), lo.sum(0), up.sum(0)])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    T = {64: traj(64), 128: traj(128)}
    t_end = time.time() + budget
    stats = dict(cases=0, esteps=0, econverge=0, failures=[], worst={k: 0.0 for k in TOL}, worst_factored=0.0, bins=0)
    seed = seed0
    one = os.environ.get("FUZZ_ONE")   # replay one seed, every metric of every E-step printed
    if one: seed = int(one); t_end = time.time() + 1e9
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n = 64 if rng.random() < 0.8 else 128
        P = T[n]
        segs = make_segments(rng, P[int(rng.integers(len(P)))])
        opts = {}
        if os.environ.get("FUZZ_DEFAULT"): rng = np.random.default_rng(seed + 10**6)   # the default plan only (the options' draws go to a stream nobody reads)
        if os.environ.get("FUZZ_BIG"): opts["chunk"] = int(rng.choice([256, 256, 512, 1024]))   # more than 4096 tiles: the two-round plan (lists A and B)
        elif rng.random() < 0.7: opts["chunk"] = int(rng.choice([37, 256, 512, 768, 1001, 1024, 2048, 3712]))
        if rng.random() < 0.5: opts["warmup"] = int(rng.choice([128, 512, 1024, 3072]))
        if rng.random() < 0.2: opts["fuse"] = 0
        if rng.random() < 0.2: opts["gap_tiles"] = 0
        if n == 64 and rng.random() < 0.4:
            opts["merge"] = 1
            if rng.random() < 0.5: opts["adapt"] = 1
            if rng.random() < 0.5: opts["prev_start"] = 1
        if os.environ.get("FUZZ_DEFAULT"): opts = {}; rng = np.random.default_rng(seed + 2 * 10**6)
        case = dict(seed=seed, n=n, segs=[len(s) for s in segs], opts=opts)
        try:
            ex = hip.HipEStep(n, mode=hip.MODE_EXACT); ex.load_segments(segs)
            fa = hip.HipEStep(n, mode=hip.MODE_FAST, **opts); fa.load_segments(segs)
            for step in range(4):
                p = P[int(rng.integers(len(P)))]
                o = ex.estep(*p)
                for kind in ("counts", "factored"):
                    try:
                        r = fa.estep(*p) if kind == "counts" else fa.estep_factored(*p)
                    except hip.HipError as err:
                        if "converge" in str(err).lower():
                            stats["econverge"] += 1; continue
                        raise
                    stats["esteps"] += 1
                    if kind == "counts":
                        m = fast_error_metrics(r, o, p[0], p[1])
                        if one:
                            Ao = np.asarray(o["A"]); Eo = np.asarray(o["E"])[:2]; Er = np.asarray(r["E"])[:2]
                            big = Eo >= 1e-6 * Eo.max(); rel = np.where(big, np.abs(Er - Eo) / np.where(big, Eo, 1.0), 0.0); w = np.unravel_index(rel.argmax(), rel.shape)
                            print("step", step, {k: float("%.3g" % v) for k, v in m.items()}, "worst E cell", w, "value %.3g of max %.3g, abs err %.3g" % (Eo[w], Eo.max(), abs(Er[w] - Eo[w])), "repairs", fa.fast_diag() if hasattr(fa, "fast_diag") else None, flush=True)
                        for k, v in m.items():
                            if v > stats["worst"][k]:
                                stats["worst"][k] = v
                                if k == "A_cell": stats["worst_A_cell_case"] = dict(case, step=step)
                        bad = {k: v for k, v in m.items() if not v <= TOL[k]}
                    else:
                        so = sums_of(np.asarray(o["A"])[:n, :n])
                        d = float(np.abs(r["sums"] - so).max() / np.abs(so).max())
                        dl = abs(r["LL"] - o["LL"]) / abs(o["LL"])
                        stats["worst_factored"] = max(stats["worst_factored"], d)
                        bad = {k: v for k, v in (("sums", d), ("LL", dl)) if not v <= (1e-10 if k == "sums" else 1e-12)}
                    if bad:
                        stats["failures"].append(dict(case, step=step, kind=kind, bad=bad))
                        print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
            ex.close(); fa.close()
        except Exception as err:                      # anything else is a failure of the case as a whole
            stats["failures"].append(dict(case, error=repr(err)[:300]))
            print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
        stats["cases"] += 1; stats["bins"] += sum(len(s) for s in segs)
        seed += 1
        if one: break
    stats["seeds"] = [seed0, seed - 1]
    print(json.dumps(stats, indent=1))


if __name__ == "__main__":
    main()
