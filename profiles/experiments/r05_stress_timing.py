#!/usr/bin/env python3
"""Round 5: what the real-data-shaped stress fixture (tests/golden/stress: N runs of 2e4 / 2e5 bins, a 5e4-bin run of homozygosity, a
het-dense stretch, a segment that is one long gap) costs the FAST E-step against a model-drawn input of the same shape: ms per
E-step, repair rounds and repaired tiles, over 10 E-steps with the parameters of one EM run moving (rd0, rd1, rd2, then rd2 kept).
-> gpurun_out/r05_stress_timing.json"""
import gzip, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psmc_amd import hip, sim
G = os.path.join(ROOT, "tests", "golden")
lut = np.full(256, 2, np.uint8); lut[ord("T")] = 0; lut[ord("K")] = 1
segs, cur = [], []
for line in gzip.open(os.path.join(G, "stress", "stress.psmcfa.gz"), "rb"):
    if line.startswith(b">"):
        if cur: segs.append(np.concatenate(cur))
        cur = []
    else: cur.append(lut[np.frombuffer(line.rstrip(b"\n"), dtype=np.uint8)])
segs.append(np.concatenate(cur))
g = dict(np.load(os.path.join(G, "stress", "stress_estep.npz")))
pars = [(g["rd%d.a" % r], g["rd%d.e" % r], g["rd%d.a0" % r]) for r in (0, 1, 2)]
plain = sim.simulate_genome(pars[0][0], pars[0][1], pars[0][2], [len(s) for s in segs], seed=5)
out = {}
opts_list = [("default", {}), ("genome_plan", dict(chunk=3712, two_phase=2, merge1=0, warm_shift=1, kc_sub=4))] + [(a, dict(kv.split("=") for kv in a.split(","))) for a in sys.argv[1:]]
for name, data in (("stress", segs), ("model_drawn_same_lengths", plain)):
    for oname, opts in opts_list:
        for factored in (0, 1):
            es = hip.HipEStep(64, mode=hip.MODE_FAST, **{k: float(v) for k, v in opts.items()})
            es.load_segments(data)
            rows = []
            for it in range(10):
                a, e, a0 = pars[min(it, 2)]
                t0 = time.perf_counter()
                (es.estep_factored if factored else es.estep)(a, e, a0)
                dt = time.perf_counter() - t0
                d = es.fast_diag(); pl = es.fast_plan()
                rows.append(dict(ms=round(dt * 1e3, 3), rounds=[d["fwd_rounds"], d["bwd_rounds"]], tiles=[d["fwd_tiles"], d["bwd_tiles"]], glued=[pl["glued_fwd"], pl["glued_bwd"]],
                                 warm_max=[pl["warm_fwd_max"], pl["warm_bwd_max"]], n_tiles=pl["tiles"], tile_len=pl["tile_len"]))
            out["%s | %s | %s" % (name, oname, "factored" if factored else "full counts")] = rows
            print(name, oname, "factored" if factored else "full", "ms", [r["ms"] for r in rows], "rounds", [r["rounds"] for r in rows][-3:], "glued", rows[-1]["glued"], flush=True)
            es.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_stress_timing.json"), "w"), indent=1)
