#!/usr/bin/env python3
"""Round 3 probes: (1) what overlaps with another wave's v_mfma_f64 on the same SIMD;
(2) where the waves of shard-sized launches land.  -> gpurun_out/r03_pipe_probe.json, r03_place_probe.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psmc_amd import hip
out_dir = os.path.join(ROOT, "gpurun_out")
kinds = ["mfma_f64", "fma_f64", "mov_dpp", "scan_levels", "ds_read_b128", "sload_readlane", "add_u32", "fma_f32", "add_f64"]
res = {"note": "cycles per round; alone = 4 waves of the kind, one per SIMD of one CU; pair = waves 0-3 matrix + waves 4-7 of the kind "
               "(one matrix wave and one wave of the kind per SIMD); serial = the pair takes the sum, overlap = the pair takes the max",
       "kinds": {}}
alone = {}
for k in kinds:
    alone[k] = hip.pipe_probe2([k] * 4)
m_alone = sum(alone["mfma_f64"]) / 4
for k in kinds:
    pair = hip.pipe_probe2(["mfma_f64"] * 4 + [k] * 4)
    two = hip.pipe_probe2([k] * 8)
    a = sum(alone[k]) / 4
    pm, pk = sum(pair[:4]) / 4, sum(pair[4:]) / 4
    longest = max(pm, pk)
    # fraction of the shorter wave's time that disappeared: 1 = perfect overlap, 0 = serial
    ov = (m_alone + a - longest) / min(m_alone, a)
    res["kinds"][k] = {"alone": round(a, 1), "two_of_kind_per_simd": round(sum(two) / 8, 1), "pair_matrix": round(pm, 1), "pair_kind": round(pk, 1),
                       "overlap_frac": round(ov, 3)}
    print("%-16s alone %7.1f  2/SIMD %7.1f | with matrix wave: matrix %7.1f kind %7.1f  overlap %.2f" % (k, a, sum(two) / 8, pm, pk, ov), flush=True)
# realistic mix: the fused step issues per 16 matrix instructions ~92 vector + 27 other; a consumer wave (matrix) beside a producer
res["matrix_alone"] = round(m_alone, 1)
json.dump(res, open(os.path.join(out_dir, "r03_pipe_probe.json"), "w"), indent=1)
pl = []
for nw in (256, 512, 768, 1024, 2048):
    for wpb in (1, 2, 4):
        for nk in (1, 2):
            r = hip.place_probe(nw, wpb, nk, 3328)
            r.update(n_waves=nw, waves_per_block=wpb, n_kernels=nk)
            pl.append(r)
            print("waves %5d x %d kernels, %d/block: %.3f ms, cycles/step mean %.0f max %.0f, SIMDs used %d (CUs %d), max waves/SIMD %d, hist %s"
                  % (nw, nk, wpb, r["ms"], r["cycles_mean"], r["cycles_max"], r["simds_used"], r["cus_used"], r["max_waves_per_simd"], r["hist"]), flush=True)
json.dump(pl, open(os.path.join(out_dir, "r03_place_probe.json"), "w"), indent=1)
