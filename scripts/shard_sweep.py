#!/usr/bin/env python3
"""E-step latency on shard-sized inputs.

Workloads: the LPT rank-0 share of the 30 M-bin benchmark genome for N = 1, 2, 4, 8 GPUs (what one rank of a
strong-scaling run holds) and the 500 k-bin single segment of config 2.  Parameters move every step
(tests/golden/traj_n64.json), as in bench.py.  Every option set given with --cfg "k=v k=v" is timed on every workload.

    python scripts/shard_sweep.py --cfg "" --cfg "chunk=512 two_phase=0" --out gpurun_out/x.json
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append", default=[])
    ap.add_argument("--shares", default="8,4,2,1", help="N of the rank-0 LPT shares to time")
    ap.add_argument("--chr", type=int, default=500000, help="bins of the config-2 segment (0 = skip)")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--factored", type=int, default=0)
    ap.add_argument("--n-states", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=1, help="go through the option sets this many times (A B A B ...): box-level drift shows up as spread between repeats")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import bench
    from psmc_amd import hip, sim
    from psmc_amd.dist import partition_segments

    a, e, a0 = bench.load_params()
    traj, _ = bench.load_trajectory(os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
    if args.n_states == 128:   # config 5: parameters of consecutive EM rounds of `psmc -N25 -p 64*2` on the benchmark genome
        from psmc_amd import hostlib
        t8 = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n128.json")))
        traj = [hostlib.hmm_params(t8["pattern"], r["params"]) for r in t8["rounds"] if r["round"] >= 1][:25]
    lens = sim.human_like_lengths(30_000_000, n_seg=90)
    full = sim.simulate_genome(a, e, a0, lens, seed=43)
    work = []
    if args.chr > 0:
        rng = np.random.default_rng(7)
        work.append(("chr22like_%dk" % (args.chr // 1000), [sim.simulate_segment(a, e, a0, args.chr, rng)]))
    for n in [int(x) for x in args.shares.split(",") if x]:
        mine = partition_segments(lens, n)[0]
        work.append(("share_1of%d" % n, [full[i] for i in mine]))
    stream = torch.cuda.current_stream()
    cfgs = (args.cfg or [""]) * max(1, args.repeat)
    res = []
    for name, segs in work:
        bins = sum(len(s) for s in segs)
        for cfg in cfgs:
            sh = bench.Shard(hip, torch, segs, args.n_states, 0, hip.MODE_FAST, cfg.split())
            es = sh.es
            try:
                if args.factored:
                    run = lambda p: es.estep_factored(*p)
                else:
                    run = lambda p: es.estep_device(p[0], p[1], p[2], sh.stats.data_ptr(), stream.cuda_stream)
                es.estep(*traj[0])
                for i in range(args.warmup):
                    run(traj[i % len(traj)])
                torch.cuda.synchronize()
                ts = []
                for i in range(args.steps):
                    t0 = time.perf_counter()
                    run(traj[(args.warmup + i) % len(traj)])
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                kern = es.timing()
                d = es.fast_diag()
                r = {"workload": name, "bins": bins, "segments": len(segs), "cfg": cfg, "ms_median": float(np.median(ts)), "ms_min": float(min(ts)),
                     "ms_max": float(max(ts)), "bins_per_s": bins / (float(np.median(ts)) * 1e-3), "tiles": d["n_chunks"], "tile_len": d["tile_len"],
                     "items": [d["items_fwd"], d["items_bwd"]], "repairs": [d["fwd_rounds"], d["bwd_rounds"], d["fwd_tiles"], d["bwd_tiles"]],
                     "kernels_ms": {k: round(float(v), 3) for k, v in kern.items()}, "plan": es.fast_plan()}
            except Exception as ex:  # one bad option set must not lose the sweep
                r = {"workload": name, "bins": bins, "cfg": cfg, "error": str(ex)}
            res.append(r)
            print(json.dumps(r), flush=True)
            sh.close()
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
