import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from psmc_amd import hip, sim
a, e, a0 = bench.load_params()
traj, _ = bench.load_trajectory(os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
def group_ms():
    g = hip.HipGroup(64, [0], mode=hip.MODE_FAST); g.load_segments(segs); g.estep(*traj[0])
    for i in range(10): g.estep(*traj[i % 25])
    t0 = time.perf_counter()
    for i in range(20): g.estep(*traj[(10 + i) % 25])
    dt = (time.perf_counter() - t0) / 20 * 1e3
    g.close(); return dt
alone = group_ms()
sh = bench.Shard(hip, torch, segs, 64, 0, hip.MODE_FAST, [])
sh.es.estep(*traj[0])
beside = group_ms()
sh2 = bench.Shard(hip, torch, segs[:30], 64, 0, hip.MODE_FAST, [])
sh2.es.estep(*traj[0])
beside2 = group_ms()
print("GPU_MAX_HW_QUEUES=%s: group alone %.3f ms, beside one idle context %.3f, beside two %.3f" % (os.environ.get("GPU_MAX_HW_QUEUES"), alone, beside, beside2))
