"""Diagnostic (GPU box): accuracy of the forward-scaled back halves per option set and EM call."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from psmc_amd import hip
import conftest, orc
g = conftest.Golden()
oracle = orc.Oracle()

def relmax(x, y):
    x = np.asarray(x); y = np.asarray(y)
    return float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300)))
def tri(A):
    lo, up = np.tril(A, -1), np.triu(A, 1)
    return np.stack([lo.sum(1), up.sum(1), np.diag(A).copy(), lo.sum(0), up.sum(0)])

for key in ("n64_curve", "n23_flat"):
    p = g.params(key); n = p["a"].shape[0]
    o = oracle.estep(p["a"], p["e"], p["a0"], g.segs_mid)
    want = tri(o["A"])
    for opts in (dict(chunk=256, warmup=512, two_phase=1), dict(chunk=256, warmup=512, two_phase=1, ckpt=0), dict(chunk=256, warmup=512, two_phase=1, learn=0),
                 dict(chunk=256, warmup=512, two_phase=1, overlap=0), dict(chunk=256, warmup=512), dict(lanes8=1), dict(lanes8=1, count_impl=0), dict()):
        es = hip.HipEStep(n, mode=hip.MODE_FAST, **opts)
        es.load_segments(g.segs_mid)
        row = []
        for it in range(3):
            r = es.estep_factored(p["a"], p["e"], p["a0"]); d = es.fast_diag()
            w = es.estep(p["a"], p["e"], p["a0"]); d2 = es.fast_diag()
            row.append("fact %.1e (rounds %d/%d items %d) full %.1e (rounds %d/%d phaseB %d)" % (relmax(r["sums"], want), d["fwd_rounds"], d["bwd_rounds"], d["items_fwd"],
                       relmax(w["A"], o["A"]), d2["fwd_rounds"], d2["bwd_rounds"], d2["phase_b_tiles"]))
        print(key, opts, " | ".join(row), flush=True)
        es.close()
