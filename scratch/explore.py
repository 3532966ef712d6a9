import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from psmc_amd import hip, sim
g = np.load("/root/repo/tests/golden/hmm_params.npz")
P = (g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"])
Ls = []
for L in sim.human_like_lengths(30_000_000, n_seg=22):
    pos = 0
    while L - pos >= 750_000: Ls.append(2000); pos += 500_000
    Ls.append((L - pos) // 250)
rng = np.random.default_rng(5)
trunks = [rng.integers(0, 2, size=l).astype(np.uint8) for l in Ls]
tot = sum(Ls)
sels = []
for r in range(100):
    s, sel = 0, []
    while s < tot:
        k = int(rng.integers(len(Ls))); sel.append(k); s += Ls[k]
    sels.append(sel)
params = [P] * 100
entries = sum(len(set(x)) for x in sels); bins = sum(sum((Ls[i] + 63) // 64 * 64 for i in set(x)) for x in sels)
print("entries", entries, "bins", bins)
for cap in (int(bins / 3.78), int(bins / 3.6)):
  for major, fill in ((1, 1), (0, 1), (1, 0), (0, 0)):
    es = hip.HipEStep(64, mode=hip.MODE_EXACT, batch_bins=cap, exact_refwd=2, batch_tailfill=fill, batch_major=major)
    es.load_segments(trunks)
    calls = []
    got = es.estep_batch(params, sels, on_done=lambda reps, out: calls.append(len(reps)))
    print("cap", cap, "major", major, "fill", fill, "launches", es.batch_info()["groups"], "callbacks", calls)
    es.close()
