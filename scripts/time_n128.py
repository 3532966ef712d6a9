"""Exact-mode timing at 128 states (BASELINE.json configs[5]-like: -p "64*2")."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psmc_amd import hip

rng = np.random.default_rng(5)
n = 128
a = rng.random((n, n)) ** 4 * 0.02 + np.eye(n) * 0.95
a /= a.sum(1, keepdims=True)
e = np.ones((3, n)); e[1] = 0.001 + rng.random(n) * 0.1; e[0] = 1 - e[1]
a0 = np.full(n, 1.0 / n)
nseg, L = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 8192
segs = [rng.choice(3, size=L, p=[0.95, 0.01, 0.04]).astype(np.uint8) for _ in range(nseg)]
es = hip.HipEStep(n, mode=hip.MODE_EXACT)
es.load_segments(segs)
es.estep(a, e, a0)
t = time.time(); r = es.estep(a, e, a0); dt = time.time() - t
print("n=128 exact: %d bins in %.3f s = %.3g bins/s; kernel ms:" % (nseg * L, dt, nseg * L / dt), es.timing())
