import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from psmc_amd import hip
g = np.load("tests/golden/hmm_params.npz"); a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
s = np.load("tests/golden/segments_mid.npz"); segs = [s[k] for k in sorted(s)]
for opts in (dict(chunk=768, warmup=256, overlap=0), dict(chunk=768, warmup=256, overlap=1), dict(chunk=768, warmup=256, overlap=2),
             dict(chunk=768, warmup=256, overlap=3), dict(chunk=2048, warmup=64, overlap=1), dict(chunk=2048, warmup=64, overlap=2)):
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); es.load_segments(segs)
    rs = []
    for i in range(8):
        r = es.estep(a, e, a0); d = es.fast_diag(); rs.append((r, d))
    base = rs[0][0]
    print(opts, [("%.1e" % (np.abs(r["A"] - base["A"]).max() / np.abs(base["A"]).max()), d["fwd_rounds"], d["bwd_rounds"], d["fwd_tiles"], d["bwd_tiles"]) for r, d in rs])
    es.close()
