#!/usr/bin/env python3
"""Random SEQUENCES of API calls on long-lived contexts, every E-step checked against the CPU oracle (round 6; companion of
scripts/fuzz_gpu.py: `python scripts/fuzz_gpu_state.py SECONDS [SEED0]`).  What a single call cannot show: state left behind --
a plan that should have been rebuilt, tables sized for the previous input, a selection that outlived its segments.

A case is one exact and one fast context (23, 64 or 128 states) and 25 random operations on BOTH: load other segments, select a
multiset of them, change plan options (fast), reserve tables (exact), an E-step with new parameters, a factored E-step (fast), a
batch of 2-5 replicates, decode / posterior of a segment (exact).  Exact results must equal the oracle's bits (the oracle is the
checker: tests/orc.py, test infrastructure); fast results are held to the suite's bounds against it."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psmc_amd import hip, hostlib, sim
from psmc_amd.parity import fast_error_metrics
import orc

TOL = dict(A_max=1e-10, E_max=1e-10, LL=1e-12, A_cell=1e-9, E_cell=1e-9, A_l1=1e-10, QA=1e-10, QE=1e-10)
PATTERN = {23: "4+5*3+4", 64: "4+25*2+4+6", 128: "64*2"}


def bits_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    return x.shape == y.shape and bool((x.view(np.uint64) == y.view(np.uint64)).all())


def params(rng, n):
    """a point of the model's parameter space near the usual start: theta, rho, max_t, free lambdas"""
    pat = PATTERN[n]
    n_free = {23: 7, 64: 28, 128: 64}[n]
    v = [np.exp(rng.normal(np.log(0.02), 0.3)), np.exp(rng.normal(np.log(0.004), 0.3)), rng.uniform(8, 20)] + list(np.exp(rng.normal(0, 0.4, size=n_free)))
    return hostlib.hmm_params(pat, v)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = orc.Oracle()
    t_end = time.time() + budget
    stats = dict(cases=0, ops={}, failures=[], worst={k: 0.0 for k in TOL})
    seed = seed0
    if os.environ.get("FUZZ_ONE"): seed = int(os.environ["FUZZ_ONE"]); t_end = time.time() + 1e9
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n = int(rng.choice([23, 64, 64, 128]))
        log = []
        def fail(what):
            stats["failures"].append(dict(seed=seed, n=n, what=what, ops=log if os.environ.get("FUZZ_ONE") else log[-8:]))
            print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
        def new_segments():
            p = params(rng, n)
            return [sim.simulate_segment(p[0], p[1], p[2], int(np.exp(rng.uniform(np.log(20), np.log(6000)))), rng, miss_rate=0.002, miss_lo=5, miss_hi=400) for _ in range(int(rng.integers(1, 8)))]
        try:
            ex = hip.HipEStep(n, mode=hip.MODE_EXACT); fa = hip.HipEStep(n, mode=hip.MODE_FAST)
            segs = new_segments(); ex.load_segments(segs); fa.load_segments(segs)
            sel = list(range(len(segs)))
            for step in range(25):
                op = str(rng.choice(["load", "select", "option", "reserve", "estep", "estep", "estep", "factored", "batch", "decode"]))
                stats["ops"][op] = stats["ops"].get(op, 0) + 1
                if op == "load":
                    segs = new_segments(); ex.load_segments(segs); fa.load_segments(segs); sel = list(range(len(segs))); log.append(("load", [len(s) for s in segs]))
                elif op == "select":
                    sel = rng.integers(0, len(segs), size=int(rng.integers(1, 2 * len(segs) + 1))).tolist(); ex.select(sel); fa.select(sel); log.append(("select", sel))
                elif op == "option":
                    k, v = [("chunk", int(rng.choice([0, 37, 256, 512, 1001, 2048]))), ("warmup", int(rng.choice([128, 512, 3072]))), ("fuse", int(rng.integers(0, 2))),
                            ("gap_tiles", int(rng.integers(0, 2))), ("merge", int(rng.integers(0, 2))), ("adapt", int(rng.integers(0, 2))), ("prev_start", int(rng.integers(0, 2)))][int(rng.integers(7))]
                    fa.set_option(k, v); log.append(("option", k, v))
                elif op == "reserve":
                    ex.reserve_tables(); log.append(("reserve",))
                elif op in ("estep", "factored"):
                    p = params(rng, n)
                    o = oracle.estep(p[0], p[1], p[2], [segs[i] for i in sel])
                    log.append((op,))
                    if op == "estep":
                        r = ex.estep(*p)
                        if not (bits_equal(r["A"], o["A"]) and bits_equal(r["E"], o["E"]) and r["LL"] == o["LL"]): fail("exact E-step differs from the oracle")
                        r = fa.estep(*p)
                        m = fast_error_metrics(r, o, p[0], p[1])
                    else:
                        r = fa.estep_factored(*p)
                        A = o["A"]; lo, up = np.tril(A, -1), np.triu(A, 1)
                        so = np.stack([lo.sum(1), up.sum(1), np.diag(A), lo.sum(0), up.sum(0)])
                        m = dict(A_max=float(np.abs(r["sums"] - so).max() / np.abs(so).max()), LL=abs(r["LL"] - o["LL"]) / abs(o["LL"]),
                                 E_max=float(np.abs(r["E"] - o["E"][:2]).max() / np.abs(o["E"][:2]).max()))
                    m["LL"] = abs(r["LL"] - o["LL"]) / max(abs(o["LL"]), 1.0)   # (a segment of missing data only has LL = 0 +- rounding: no relative error of that)
                    for k, v in m.items(): stats["worst"][k] = max(stats["worst"][k], v)
                    bad = {k: v for k, v in m.items() if not v <= TOL[k]}
                    if bad: fail("fast %s out of bounds: %s (LL %r against %r)" % (op, bad, r["LL"], o["LL"]))
                elif op == "batch":
                    R = int(rng.integers(2, 6))
                    sels = [rng.integers(0, len(segs), size=int(rng.integers(1, len(segs) + 2))).tolist() for _ in range(R)]
                    ps = [params(rng, n) for _ in range(R)]
                    got = ex.estep_batch(ps, sels); gf = fa.estep_batch(ps, sels)
                    log.append(("batch", R))
                    for r in range(R):
                        o = oracle.estep(ps[r][0], ps[r][1], ps[r][2], [segs[i] for i in sels[r]])
                        if not (bits_equal(got["A"][r], o["A"]) and bits_equal(got["E"][r], o["E"]) and got["LL"][r] == o["LL"]): fail("exact batch replicate %d differs from the oracle" % r)
                        m = fast_error_metrics(dict(A=gf["A"][r], E=gf["E"][r], LL=gf["LL"][r]), o, ps[r][0], ps[r][1])
                        m["LL"] = abs(gf["LL"][r] - o["LL"]) / max(abs(o["LL"]), 1.0)
                        bad = {k: v for k, v in m.items() if not v <= TOL[k]}
                        if bad: fail("fast batch replicate %d out of bounds: %s (LL %r against %r; selection %s)" % (r, bad, gf["LL"][r], o["LL"], sels[r]))
                    ex.select(sel); fa.select(sel)   # (a batch leaves the context's selection as it was? make it explicit either way)
                elif op == "decode":
                    p = params(rng, n); k = int(rng.integers(len(segs)))
                    ex.select(list(range(len(segs)))); ex.estep(*p)
                    f, b, s, lk, chk = oracle.fwd_bwd(p[0], p[1], p[2], segs[k])
                    path, mp = oracle.post_decode(f, b, s); gp, gm = ex.decode(k)
                    post, rec = oracle.post_full(p[0], p[1], segs[k], f, b, s); pp, rr = ex.posterior(k)
                    log.append(("decode", k))
                    if not (np.array_equal(gp, path[1:]) and bits_equal(gm, mp[1:]) and bits_equal(pp, post[1:]) and bits_equal(rr, rec[1:])): fail("decode / posterior differs from the oracle")
                    ex.select(sel)
            ex.close(); fa.close()
        except Exception as err:
            fail("exception: " + repr(err)[:300])
        stats["cases"] += 1
        seed += 1
        if os.environ.get("FUZZ_ONE"): break
    stats["seeds"] = [seed0, seed - 1]
    print(json.dumps(stats, indent=1))


if __name__ == "__main__":
    main()
