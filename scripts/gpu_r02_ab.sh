#!/bin/bash
# round 2, last seconds of GPU: default path after the kc_warm plumbing; kc_warm=1 parity and timing (experimental)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_estep.py -m gpu -q -x -k "learns or stress_tiny or factored_statistics" 2>&1 | tail -1
timeout 30 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-300
import sys, numpy as np
sys.path.insert(0, "tests")
from psmc_amd import hip
import conftest, orc
g = conftest.Golden(); oracle = orc.Oracle(); p = g.params("n64_curve")
o = oracle.estep(p["a"], p["e"], p["a0"], g.segs_mid)
rel = lambda x, y: float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300)))
for kw in (dict(kc_warm=1), dict(kc_warm=0)):
    es = hip.HipEStep(64, mode=hip.MODE_FAST, chunk=768, warmup=256, group_cap=200000, **kw); es.load_segments(g.segs_mid)
    out = []
    for it in range(6):
        r = es.estep(p["a"], p["e"], p["a0"]); d = es.fast_diag()
        out.append("%.1e r%d/%d i%d/%d" % (rel(r["A"], o["A"]), d["fwd_rounds"], d["bwd_rounds"], d["items_fwd"], d["items_bwd"]))
    print(kw, out, flush=True); es.close()
PY
timeout 60 python bench.py --steps 10 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 --opt kc_warm=1 > gpurun_out/ab_bench.json 2> gpurun_out/ab_bench.err
python - <<PY
import json
r=json.load(open("gpurun_out/ab_bench.json")); k=r["roofline"]["kernels_ms"]; fk=r["factored_stats"].get("kernels_ms") or {}
print("kc_warm=1: moving %.2f factored %.2f | fwd %.2f expect %.2f | factored fwd %.2f acc %.2f total %.2f repairs %s" % (r["ms_per_step"], r["factored_stats"]["ms_per_step"], k["fwd_sweep"], k["expect"], fk.get("fwd_sweep",0), fk.get("expect",0), fk.get("total",0), r["config"].get("repair_rounds")))
PY
