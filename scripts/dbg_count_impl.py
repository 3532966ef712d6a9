"""Diagnostic (GPU box): run-to-run reproducibility of the fast E-step per back-half variant.
Prints max |r1 - r2| of two consecutive E-steps on one context and of two fresh contexts."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from psmc_amd import hip
import conftest
g = conftest.Golden()
p = g.params("n64_curve")

def d(x, y): return float(np.max(np.abs(np.asarray(x) - np.asarray(y))))

for kw in (dict(count_impl=0), dict(count_impl=1), dict(count_impl=2)):
    for extra in (dict(chunk=512, learn=0), dict(chunk=512, learn=0, two_phase=0), dict(chunk=512, learn=0, overlap=0)):
        es = hip.HipEStep(64, mode=hip.MODE_FAST, **kw, **extra)
        es.load_segments(g.segs_mid)
        r = [es.estep(p["a"], p["e"], p["a0"]) for _ in range(3)]
        dg = es.fast_diag()
        f = [es.estep_factored(p["a"], p["e"], p["a0"]) for _ in range(3)]
        print(kw, extra, "A", d(r[0]["A"], r[1]["A"]), d(r[1]["A"], r[2]["A"]), "E", d(r[0]["E"], r[1]["E"]), "LL", r[0]["LL"] - r[1]["LL"],
              "rounds", dg["fwd_rounds"], dg["bwd_rounds"], "| factored sums", d(f[0]["sums"], f[1]["sums"]), d(f[1]["sums"], f[2]["sums"]),
              "E", d(f[0]["E"], f[1]["E"]), flush=True)
        es.close()
