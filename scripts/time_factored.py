"""Fast mode, factored statistics (psmc_hip_estep_factored) vs the full-matrix E-step on the benchmark genome."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip, sim
g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
es = hip.HipEStep(64, mode=hip.MODE_FAST, **{k: float(v) for k, v in (kv.split("=") for kv in sys.argv[1:])})
es.load_segments(segs)
for name, fn in (("factored", es.estep_factored), ("full A", es.estep)):
    for it in range(5):
        t = time.perf_counter(); r = fn(a, e, a0); dt = time.perf_counter() - t
        d = es.fast_diag()
        print("%-8s call %d: %.1f ms = %.3g bins/s; rounds %d/%d" % (name, it, dt * 1e3, int(lens.sum()) / dt, d["fwd_rounds"], d["bwd_rounds"]))
    print("   ", {k: round(float(v), 2) for k, v in es.timing().items()}, "LL", r["LL"])
