"""128 states (-p "64*2", BASELINE.json configs[5]): exact-mode timing on many short segments and fast-mode
timing on a genome-sized batch (golden n128 parameters, simulated observations)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip, sim

g = np.load(os.path.join(ROOT, "tests", "golden", "estep_n128.npz"))
a, e, a0 = g["n128_curve.a"], g["n128_curve.e"], g["n128_curve.a0"]
what = sys.argv[1] if len(sys.argv) > 1 else "fast"
if what == "exact":
    rng = np.random.default_rng(5)
    nseg, L = int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    segs = [rng.choice(3, size=L, p=[0.95, 0.01, 0.04]).astype(np.uint8) for _ in range(nseg)]
    es = hip.HipEStep(128, mode=hip.MODE_EXACT)
    es.load_segments(segs)
    es.estep(a, e, a0)
    t = time.time(); r = es.estep(a, e, a0); dt = time.time() - t
    print("n=128 exact: %d bins in %.3f s = %.3g bins/s; kernel ms:" % (nseg * L, dt, nseg * L / dt), es.timing())
else:
    bins = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000_000
    lens = sim.human_like_lengths(bins, n_seg=90)
    segs = sim.simulate_genome(a, e, a0, lens, seed=43)
    es = hip.HipEStep(128, mode=hip.MODE_FAST)
    es.load_segments(segs)
    for it in range(5):
        t = time.time(); r = es.estep(a, e, a0); dt = time.time() - t
        d = es.fast_diag()
        print("n=128 fast call %d: %d bins in %.1f ms = %.3g bins/s; rounds %d/%d items %d/%d" % (
            it, int(lens.sum()), dt * 1e3, int(lens.sum()) / dt, d["fwd_rounds"], d["bwd_rounds"], d["items_fwd"], d["items_bwd"]))
    print({k: round(float(v), 2) for k, v in es.timing().items()}, "LL", r["LL"])
