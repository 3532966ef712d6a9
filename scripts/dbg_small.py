import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from psmc_amd import hip
g = np.load("tests/golden/hmm_params.npz"); a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
s = np.load("tests/golden/segments_small.npz"); segs = [s[k] for k in sorted(s)]
seg = segs[11]
fa = hip.HipEStep(64, mode=hip.MODE_FAST, overlap=0); fa.load_segments([seg])
f = fa.estep(a, e, a0)
X, bt, inv = fa.tables(0)
L = len(seg)
d = np.ones(L); pos = np.arange(1, L + 1); m = (pos % 4 == 0); d[m] = 1.0 / inv[m]
post = (X * bt * d[:, None]).sum(1) / 1.0   # gamma_p summed over k (e=1 for missing)
print("posterior sums (should be 1):", post[[0,1,2,100,254,255,256,257,258,297,298]])
xi = np.array([(X[p][:, None] * a * bt[p + 1][None, :]).sum() for p in range(L - 1)])
print("xi sums:", xi[[0,1,100,253,254,255,256,257,297,298]])
print("bt rows 255..258 first vals", bt[254:259, :3])
print("X sums", X.sum(1)[[0,1,3,4,255,256,257]])
