import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from psmc_amd import hip
import conftest, orc
orc.build_oracle(); g = conftest.Golden(); o_ = orc.Oracle()
p = g.params("n64_curve")
segs = g.segs_small + g.segs_mid[3:]
def relmax(x, y): return float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
opts = {k: float(v) for k, v in (kv.split("=") for kv in sys.argv[1].split(",") if kv)}
for i, sg in enumerate(segs):
    o = o_.estep(p["a"], p["e"], p["a0"], [sg])
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); es.load_segments([sg])
    r = es.estep(p["a"], p["e"], p["a0"]); d = es.fast_diag()
    dA = np.abs(r["A"] - o["A"]); k, l = np.unravel_index(np.argmax(dA), dA.shape)
    print("seg %2d L=%6d: A %.1e E %.1e  rounds %d/%d tiles %d  worst cell (%d,%d) got %.6g want %.6g  sumA got %.10g want %.10g" % (
        i, len(sg), relmax(r["A"], o["A"]), relmax(r["E"], o["E"]), d["fwd_rounds"], d["bwd_rounds"], d["n_chunks"], k, l, r["A"][k, l], o["A"][k, l], r["A"].sum(), o["A"].sum()))
    es.close()
