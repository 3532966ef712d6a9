#!/usr/bin/env python3
"""Per-E-step trace over 30 moving-parameter E-steps: ms, repair rounds / tiles, the plan (mean warm-ups, glued tiles); 30 M-bin genome (share 1)
or rank 0's share of N GPUs.  (Written for the adaptive warm-ups of round 3, profiles/r03_adaptive_warmup_trace.txt; round 6 built them again on top of merging repairs.)"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from psmc_amd import hip, sim
from psmc_amd.dist import partition_segments
share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opts = sys.argv[2].split() if len(sys.argv) > 2 else []
a, e, a0 = bench.load_params()
traj, _ = bench.load_trajectory(os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
lens = sim.human_like_lengths(30_000_000, n_seg=90)
full = sim.simulate_genome(a, e, a0, lens, seed=43)
segs = [full[i] for i in partition_segments(lens, share)[0]]
sh = bench.Shard(hip, torch, segs, 64, 0, hip.MODE_FAST, opts)
stream = torch.cuda.current_stream()
for it in range(30):
    p = traj[it % len(traj)]
    t0 = time.perf_counter()
    sh.es.estep_device(p[0], p[1], p[2], sh.stats.data_ptr(), stream.cuda_stream); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    d = sh.es.fast_diag(); pl = sh.es.fast_plan(); k = sh.es.timing()
    print("step %2d  %7.3f ms  repairs %d/%d rounds %d/%d tiles (%d merged, recount %d) | warm f %4.0f b %4.0f max %d/%d glued %d/%d | fwd %.2f cnt %.2f" % (it, ms, d["fwd_rounds"], d["bwd_rounds"], d["fwd_tiles"], d["bwd_tiles"], d["merged"], d["recounted"],
          pl["warm_fwd_mean"], pl["warm_bwd_mean"], pl["warm_fwd_max"], pl["warm_bwd_max"], pl["glued_fwd"], pl["glued_bwd"], k["fwd_sweep"], k["expect"]), flush=True)
