import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from psmc_amd import hip, sim
bins = 30_000_000
g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
lens = sim.human_like_lengths(bins, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
for opts in (dict(struct_tiles=4096, warmup=2048), dict(struct_tiles=4096, warmup=2048, learn=0), dict()):
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts)
    t0 = time.perf_counter(); es.load_segments(segs); t1 = time.perf_counter()
    print(opts, "load %.1f ms" % ((t1 - t0) * 1e3))
    for i in range(3):
        t0 = time.perf_counter(); r = es.estep(a, e, a0); dt = time.perf_counter() - t0
        print("  call %d: %.1f ms" % (i, dt * 1e3), es.timing(), {k: v for k, v in es.fast_diag().items() if "round" in k or "tiles" in k or "items" in k})
    es.close()
