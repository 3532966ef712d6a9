"""Option sweep of the factored E-step on the benchmark genome: one line per option set (min / median ms of 6 calls after 3 learning calls)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip, sim
g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
SETS = [s for s in sys.argv[1:]] or [""]
full = "--full" in SETS
SETS = [s for s in SETS if s != "--full"]
for spec in SETS:
    opts = {k: float(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts)
    es.load_segments(segs)
    fn = es.estep if full else es.estep_factored
    ts = []
    for it in range(9):
        t = time.perf_counter(); r = fn(a, e, a0); dt = time.perf_counter() - t
        if it >= 3: ts.append(dt * 1e3)
    d = es.fast_diag(); tm = es.timing()
    print("%-44s min %.2f med %.2f ms | tiles %d x %d, rounds %d/%d | fwd %.1f acc %.1f | LL %.6f" % (
        spec or "(default)", min(ts), float(np.median(ts)), d["n_chunks"], d["tile_len"], d["fwd_rounds"], d["bwd_rounds"],
        tm.get("fwd_sweep", 0), tm.get("expect", 0), r["LL"]), flush=True)
    es.close()
