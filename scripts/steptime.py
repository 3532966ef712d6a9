"""Per-step latency of the fast sweeps in isolation: one segment, one tile (one wave)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psmc_amd import hip, sim
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hmm_params.npz"))
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
L = 262144
rng = np.random.default_rng(3)
for nseg in (1, 4, 16):
    segs = [sim.simulate_segment(a, e, a0, L, rng) for _ in range(nseg)]
    for st in (1, 0):
        es = hip.HipEStep(64, mode=hip.MODE_FAST, structured=st, chunk=L, overlap=0)
        es.load_segments(segs)
        es.estep(a, e, a0); r = es.estep(a, e, a0)
        t = es.timing(); d = es.fast_diag()
        print("nseg %2d %s tiles %d: fwd %.3f ms = %.1f ns/step, bwd %.3f ms = %.1f ns/step, expect %.3f ms" % (
            nseg, "struct" if st else "dense ", d["n_chunks"], t["fwd_sweep"], t["fwd_sweep"] * 1e6 / L, t["bwd_sweep"], t["bwd_sweep"] * 1e6 / L, t["expect"]))
        es.close()
