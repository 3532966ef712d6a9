#!/usr/bin/env python3
"""Config 5 alone (30 M bins x 128 states, fast mode) for the profilers: N full-count E-steps, then N factored ones."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from psmc_amd import hip, sim, hostlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
a, e, a0 = bench.load_params()
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
tj = os.path.join(ROOT, "tests", "golden", "traj_n128.json")
if os.path.exists(tj):
    t = json.load(open(tj)); mov = [hostlib.hmm_params(t["pattern"], r["params"]) for r in t["rounds"] if r["round"] >= 1][:25]
else:
    g = np.load(os.path.join(ROOT, "tests", "golden", "estep_n128.npz")); mov = [(g["n128_curve.a"], g["n128_curve.e"], g["n128_curve.a0"])]
sh = bench.Shard(hip, torch, segs, 128, 0, hip.MODE_FAST, [])
st = torch.zeros(128 * 128 + 2 * 128 + 1, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream()
sh.es.estep(*mov[0])
for i in range(n):
    sh.es.estep_device(*mov[i % len(mov)], st.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
for i in range(n):
    sh.es.estep_factored(*mov[i % len(mov)])
print(json.dumps({"bins": sh.bins, "n": n, "kernels_ms": sh.es.timing()}))
