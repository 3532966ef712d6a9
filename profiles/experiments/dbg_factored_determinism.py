#!/usr/bin/env python3
"""Are the factored statistics bit-reproducible between two contexts with the same call history?  And between a batch
child (shared tables) and a fresh context?  Prints the first difference."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psmc_amd import hip, hostlib

tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
params = [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"][1:5]]
s = np.load(os.path.join(ROOT, "tests", "golden", "segments_mid.npz")); segs = [s[k] for k in sorted(s)]
sels = [[5, 4, 5, 3, 5], [0, 1, 2], [2, 2, 1, 0, 4], list(range(6))]


def beq(x, y):
    return np.array_equal(np.ascontiguousarray(x).view(np.uint64), np.ascontiguousarray(y).view(np.uint64))


base = dict(chunk=768, warmup=256, group_cap=200000)
variants = [base, dict(base, ckpt=0), dict(base, learn=0), dict(base, kc_min=0), dict(base, merge1=0), dict(base, two_phase=2), dict(base, overlap=0), dict(base, fuse=0), dict()]
if len(sys.argv) > 1:
    variants = variants[:int(sys.argv[1])]
for opts in variants:
    print("== opts", opts)
    # (1) two fresh contexts, same history: estep, factored, estep, factored ...
    ctx = []
    for rep in range(2):
        f = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); f.load_segments(segs); f.select(sels[0]); ctx.append(f)
    for it in range(3):
        r = [(c.estep(*params[0]), c.estep_factored(*params[0]), c.fast_diag()) for c in ctx]
        print(" fresh-vs-fresh it", it, "A", beq(r[0][0]["A"], r[1][0]["A"]), "sums", beq(r[0][1]["sums"], r[1][1]["sums"]), "E", beq(r[0][1]["E"], r[1][1]["E"]),
              "LL", r[0][1]["LL"] == r[1][1]["LL"], "maxdiff sums %.3e" % np.abs(r[0][1]["sums"] - r[1][1]["sums"]).max(),
              "repairs", [(d["fwd_rounds"], d["bwd_rounds"], d["fwd_tiles"], d["bwd_tiles"]) for _, _, d in r])
    # (2) factored only, twice in a row on one context, learn as given
    f = ctx[0]
    a = f.estep_factored(*params[0]); b = f.estep_factored(*params[0])
    print(" same ctx twice: sums", beq(a["sums"], b["sums"]), "%.3e" % np.abs(a["sums"] - b["sums"]).max())
    for c in ctx:
        c.close()
    # (3) batch children vs fresh
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); es.load_segments(segs)
    fresh = []
    for sel in sels:
        f = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); f.load_segments(segs); f.select(sel); fresh.append(f)
    for it in range(3):
        got = es.estep_batch(params, sels, want="both")
        for r in range(4):
            w = fresh[r].estep(*params[r]); wf = fresh[r].estep_factored(*params[r])
            print(" batch-vs-fresh it", it, "rep", r, "A", beq(got["A"][r], w["A"]), "sums", beq(got["sums"][r], wf["sums"]), "E", beq(got["E"][r], wf["E"]),
                  "LL", got["LL"][r] == wf["LL"], "maxdiff %.3e" % np.abs(got["sums"][r] - wf["sums"]).max())
    for f in fresh:
        f.close()
    es.close()
