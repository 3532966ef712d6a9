import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from psmc_amd import hip
import conftest, orc
orc.build_oracle(); g = conftest.Golden(); o_ = orc.Oracle()
p = g.params("n64_curve")
segs = g.segs_small + g.segs_mid[3:]
o = o_.estep(p["a"], p["e"], p["a0"], segs)
def relmax(x, y): return float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
for spec in sys.argv[1:]:
    opts = {k: float(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts); es.load_segments(segs)
    res = []
    for it in range(4):
        r = es.estep(p["a"], p["e"], p["a0"]); d = es.fast_diag()
        res.append("%.1e/%.1e r%d/%d i%d" % (relmax(r["A"], o["A"]), relmax(r["E"], o["E"]), d["fwd_rounds"], d["bwd_rounds"], d["items_fwd"]))
    print("%-40s" % spec, " | ".join(res), flush=True)
    es.close()
