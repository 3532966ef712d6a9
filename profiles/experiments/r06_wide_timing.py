#!/usr/bin/env python3
"""Round 6 (the matrix in registers at 129 .. 208 states; round 5: profiles/r05_wide_timing.json): what the wide exact kernels (129 .. 1024 states, estep_wide.hip) cost -- an exact E-step at 149, 200, 256 and 512 states
over 20 segments of 50,000 bins (HIP events per kernel), with the oracle's time on one host core for a slice beside it.
-> gpurun_out/r06_wide_timing.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psmc_amd import hip
import orc
rng = np.random.default_rng(5)
out = {}
for n in (149, 192, 200, 208, 224, 256):
    a = rng.random((n, n)) ** 4 * 0.02 + np.eye(n) * (0.9 + 0.1 * rng.random(n)); a /= a.sum(1, keepdims=True)
    e = np.ones((3, n)); e[1] = 0.001 + rng.random(n) * 0.15; e[0] = 1.0 - e[1]
    a0 = rng.random(n) + 0.1; a0 /= a0.sum()
    segs = [rng.choice(3, size=50000, p=[0.9, 0.07, 0.03]).astype(np.uint8) for _ in range(20)]
    es = hip.HipEStep(n, mode=hip.MODE_EXACT)
    es.load_segments(segs)
    es.estep(a, e, a0)
    t0 = time.perf_counter(); r = es.estep(a, e, a0); dt = time.perf_counter() - t0
    k = es.timing()
    t1 = time.perf_counter(); o = orc.Oracle().estep(a, e, a0, [segs[0][:5000]]); dto = time.perf_counter() - t1
    es.select([0]); r1 = es.estep(a, e, a0)
    out[n] = dict(bins=1_000_000, ms=dt * 1e3, kernels_ms={q: float(v) for q, v in k.items()}, us_per_bin_of_a_segment=k["forward"] * 1e3 / 50000 + k["backward"] * 1e3 / 50000,
                  bins_per_s=1e6 / dt, oracle_one_core_bins_per_s=5000 / dto)
    print(n, json.dumps(out[n]), flush=True)
    es.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_wide_timing.json"), "w"), indent=1)
