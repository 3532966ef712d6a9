#!/usr/bin/env python3
"""SQ-counter pass of scripts/lease.sh prof -> profiles/<tag>_<name>_sq_counters.json.

    python scripts/sq_summary.py r03 n64      # reads gpurun_out/pmc/n64_SQ_{counter_collection,kernel_trace}.csv

Per kernel: the counters of its LONGEST launch (a steady-state launch over the whole genome; the short first-call and
repair launches would blur the ratios) and the sums over every launch of the E-step that contains it.  Counter
collection serialises the kernels, so the times here are not the pipeline's; instruction counts and the ratios between
counters are what this file is for (the SQ block sees about 0.8 of the device: profiles/r02_sq_counters.json)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """psmc::k_fwd_struct<false, 4, true>(double const*, ...) -> k_fwd_struct<false,4,true>"""
    m = re.search(r"psmc::(k_[a-z0-9_]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name.split("(")[0]


def main():
    tag, name = sys.argv[1], sys.argv[2]
    base = os.path.join(ROOT, "gpurun_out", "pmc", name + "_SQ")
    per = collections.defaultdict(dict)   # dispatch -> counter -> value
    meta = {}
    for r in csv.DictReader(open(base + "_counter_collection.csv")):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, int(r["Grid_Size"]))
    kernels = collections.defaultdict(list)
    for d, (k, ms, grid) in meta.items():
        if k.startswith("k_"):
            kernels[k].append((ms, d, grid))
    out = {}
    for k, ls in sorted(kernels.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        ms, d, grid = max(ls)
        e = {"launches": len(ls), "longest_ms": round(ms, 3), "grid_threads": grid}
        e.update({c: v for c, v in sorted(per[d].items())})
        tot = collections.defaultdict(float)
        for _, dd, _ in ls:
            for c, v in per[dd].items():
                tot[c] += v
        e["sum_over_launches"] = {c: v for c, v in sorted(tot.items())}
        if e.get("SQ_INSTS_MFMA", 0) > 0:
            e["valu_per_mfma"] = e["SQ_INSTS_VALU"] / e["SQ_INSTS_MFMA"]
            e["busy_cycles_per_mfma"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_INSTS_MFMA"] if "SQ_VALU_MFMA_BUSY_CYCLES" in e else None
        if e.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_INSTS_VALU" in e:
            e["valu_per_wave_cycle"] = e["SQ_INSTS_VALU"] / e["SQ_WAVE_CYCLES"]
        out[k] = e
    cmd = open(base.replace("_SQ", "_SQ.cmd")).read().strip() if os.path.exists(base.replace("_SQ", "_SQ.cmd")) else ""
    res = {"command": cmd, "note": __doc__.split("\n\n", 2)[2].replace("\n", " "), "kernels": out}
    dst = os.path.join(ROOT, "profiles", "%s_%s_sq_counters.json" % (tag, name))
    json.dump(res, open(dst, "w"), indent=1)
    tv = sum(e.get("SQ_INSTS_VALU", 0) for e in out.values())
    for k, e in out.items():
        print("%-34s %3d launches  longest %8.3f ms  VALU %.3e (%4.1f %%)  MFMA %.3e  LDS %.3e  SALU %.3e" % (
            k, e["launches"], e["longest_ms"], e.get("SQ_INSTS_VALU", 0), 100 * e.get("SQ_INSTS_VALU", 0) / max(tv, 1), e.get("SQ_INSTS_MFMA", 0), e.get("SQ_INSTS_LDS", 0), e.get("SQ_INSTS_SALU", 0)))
    print("->", dst)


if __name__ == "__main__":
    main()
