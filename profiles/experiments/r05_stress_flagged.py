import gzip, os, sys
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT)
from psmc_amd import hip
G = os.path.join(ROOT, "tests", "golden")
lut = np.full(256, 2, np.uint8); lut[ord("T")] = 0; lut[ord("K")] = 1
segs, cur = [], []
for line in gzip.open(os.path.join(G, "stress", "stress.psmcfa.gz"), "rb"):
    if line.startswith(b">"):
        if cur: segs.append(np.concatenate(cur))
        cur = []
    else: cur.append(lut[np.frombuffer(line.rstrip(b"\n"), dtype=np.uint8)])
segs.append(np.concatenate(cur))
g = dict(np.load(os.path.join(G, "stress", "stress_estep.npz")))
a, e, a0 = g["rd2.a"], g["rd2.e"], g["rd2.a0"]
es = hip.HipEStep(64, mode=hip.MODE_FAST, chunk=3712, two_phase=2, merge1=0, warm_shift=1, kc_sub=4)
es.load_segments(segs)
for it in range(6):
    if it == 5: os.environ["PSMC_HIP_DEBUG_FLAGGED"] = "1"
    es.estep(a, e, a0)
print(es.fast_diag())
# how stationary is a0 for a, and how fast does the chain forget in a gap
pi = a0.copy()
for _ in range(200000): pi = pi @ a
print("|a0 - pi| / |pi| =", np.abs(a0 - pi).max() / pi.max(), " |a0 a - a0| =", np.abs(a0 @ a - a0).max())
w = np.linalg.eigvals(a); w = np.sort(np.abs(w))[::-1]; print("eigenvalues", w[:4], "bins to 1e-12:", np.log(1e-12) / np.log(w[1]))
