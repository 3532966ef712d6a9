#!/usr/bin/env python3
"""Round 5: bench.py's `real_shape` extra (the 30 M-bin benchmark genome with 1.0 M bins of planted centromere / telomere gaps) with
gap_tiles = 1 (default) and 0 (rounds 1-4), full counts, parameters moving.  -> gpurun_out/r05_real_shape_ab.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from psmc_amd import hip, sim
a, e, a0 = bench.load_params()
moving, _ = bench.load_trajectory(os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
stream = torch.cuda.Stream()
out = {}
for name, opts in (("gap_tiles=1", []), ("gap_tiles=0", ["gap_tiles=0"])):
    with torch.cuda.stream(stream):
        out[name] = bench.real_shape_extra(hip, torch, segs, moving, 0, opts, stream, 12.3)
    print(name, {k: v for k, v in out[name].items() if k not in ("workload", "kernels_ms")}, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_real_shape_ab.json"), "w"), indent=1)
