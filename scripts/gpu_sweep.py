#!/usr/bin/env python3
"""Fast-mode tuning sweep on one GPU: tile length x speculative overlap x waves/tile,
one synthetic genome generated once.  Prints one line per configuration."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from psmc_amd import hip, sim

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
lens = sim.human_like_lengths(bins, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
off = np.concatenate([[0], np.cumsum((lens.astype(np.int64) + 63) // 64 * 64)])
host = np.full(int(off[-1]) + 256, 2, dtype=np.uint8)
for s, o in zip(segs, off[:-1]):
    host[o:o + len(s)] = s
d_obs = torch.from_numpy(host).cuda()
ref = None
grid = [dict(structured=0), dict(), dict(overlap=0), dict(struct_tiles=2048), dict(struct_tiles=4096), dict(struct_tiles=16384),
        dict(struct_tiles=32768), dict(struct_tiles=4096, warmup=2048), dict(struct_tiles=8192, warmup=2048),
        dict(struct_tiles=16384, warmup=2048), dict(struct_tiles=16384, warmup=1024), dict(struct_tiles=8192, warmup=8192),
        dict(struct_tiles=8192, n_sub=2), dict(struct_tiles=4096, n_sub=3)]
if len(sys.argv) > 2:
    grid = [json.loads(x) for x in sys.argv[2:]]
for opts in grid:
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts)
    es.load_segments_device(d_obs.data_ptr(), off[:-1], lens, keepalive=d_obs)
    hist = []
    for _ in range(4):
        t0 = time.perf_counter(); r = es.estep(a, e, a0); hist.append(round((time.perf_counter() - t0) * 1e3, 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = es.estep(a, e, a0)
    dt = (time.perf_counter() - t0) / 3
    t = es.timing(); d = es.fast_diag()
    if ref is None:
        ref = r
    err = float(np.abs(r["A"] - ref["A"]).max() / np.abs(ref["A"]).max())
    print(json.dumps(dict(opts=opts, ms=round(dt * 1e3, 2), bins_per_s=round(int(lens.sum()) / dt / 1e6, 1),
                          chains=round(t["forward"], 2), tail=round(t["backward"], 2), exp=round(t["expect"], 2), fsw=round(t["fwd_sweep"], 2), bsw=round(t["bwd_sweep"], 2),
                          red=round(t["reduce"], 2), tiles=d["n_chunks"], rounds=[d["fwd_rounds"], d["bwd_rounds"]],
                          rep_tiles=[d["fwd_tiles"], d["bwd_tiles"]], struct=d["structured"], tile=d["tile_len"], items=[d["items_fwd"], d["items_bwd"]], hist=hist, dA_vs_first=err, LL=r["LL"])), flush=True)
    es.close()
