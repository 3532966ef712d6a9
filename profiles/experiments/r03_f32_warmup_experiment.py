#!/usr/bin/env python3
"""CPU experiment (round 3): the first part of a tile's warm-up in FP32.

Dense numpy recursion on a simulated 400 k-bin segment with the parameters of three EM rounds (tests/golden/traj_n64.json).  For
120 tile starts: the mismatch of the speculated start vector against the exact one (FP64 from the segment's true start) after
a warm-up of W = 3072 bins from the stationary vector -- all FP64, and with the first (1 - phi) W steps in float32 (vector,
matrix and emission rows rounded to float32, float32 products and sums) followed by phi W steps in FP64."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from psmc_amd import sim

traj, _ = bench.load_trajectory(os.path.join(ROOT, 'tests', 'golden', 'traj_n64.json'))
a, e, a0 = bench.load_params()
L, W = 400000, 3072
seq = sim.simulate_segment(a, e, a0, L, np.random.default_rng(11))
pts = list(range(3 * W + 100, L, (L - 3 * W) // 120))[:120]


def run(par, x, lo, hi, dt):
    A = par[0].astype(dt); E3 = np.vstack([par[1][:2], np.ones(64)]).astype(dt)
    x = x.astype(dt)
    for p in range(lo, hi + 1):
        x = E3[seq[p]] * (x @ A)
        if p % 4 == 0: x = x / x.sum()
    return x / x.sum()


for r in (1, 12, 24):
    par = traj[r]
    A, E, A0 = par
    # exact vectors at the sampled positions: FP64 from far enough upstream (3 W)
    errs = {}
    for p in pts:
        ex = run(par, A0, p - 3 * W, p - 1, np.float64)
        x64 = run(par, A0, p - W, p - 1, np.float64)
        errs.setdefault('fp64', []).append(np.abs(x64 - ex).max() / ex.max())
        for phi in (0.5, 0.57, 0.65, 0.75):
            n64 = int(round(phi * W / 16)) * 16
            xh = run(par, A0, p - W, p - n64 - 1, np.float32).astype(np.float64)
            xh = run(par, xh, p - n64, p - 1, np.float64)
            errs.setdefault('fp32 then %.2f W in fp64' % phi, []).append(np.abs(xh - ex).max() / ex.max())
    print("EM round %d:" % r)
    for k, v in errs.items():
        v = np.array(v)
        print("   %-28s mismatch median %.1e  p90 %.1e  max %.1e   above 1e-12: %3d of %d" % (k, np.median(v), np.percentile(v, 90), v.max(), (v > 1e-12).sum(), len(v)))
