"""Structured vs dense fast sweeps vs oracle on golden params (debug)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from psmc_amd import hip
import orc
from conftest import Golden
g = Golden()
orc.build_oracle()
O = orc.Oracle()
print("selftest", hip.selftest(0))
def rel(x, y): return float(np.abs(np.asarray(x) - np.asarray(y)).max() / np.abs(np.asarray(y)).max())
for key in ("n64_curve", "n64_flat", "n23_curve"):
    p = g.params(key)
    n = p["a"].shape[0]
    for segs, nm in ((g.segs_small, "small"), (g.segs_mid, "mid")):
        o = O.estep(p["a"], p["e"], p["a0"], segs)
        for st in (2, 1):
            for opts in (dict(), dict(chunk=256, warmup=512), dict(chunk=1024, warmup=64), dict(chunk=1000, warmup=100, overlap=0)):
                es = hip.HipEStep(n, mode=hip.MODE_FAST, structured=1, fuse=st - 1, **opts)
                es.load_segments(segs)
                r = es.estep(p["a"], p["e"], p["a0"])
                r = es.estep(p["a"], p["e"], p["a0"])
                d = es.fast_diag()
                print(key, nm, "fused " if st == 2 else "struct", opts, "A %.2e E %.2e LL %.2e" % (rel(r["A"], o["A"]), rel(r["E"], o["E"]), abs(r["LL"] - o["LL"]) / abs(o["LL"])),
                      "used_struct", d["structured"], "tiles", d["n_chunks"], "rounds", d["fwd_rounds"], d["bwd_rounds"], "nrep", d["fwd_tiles"], d["bwd_tiles"])
                es.close()
