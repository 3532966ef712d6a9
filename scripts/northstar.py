#!/usr/bin/env python3
"""The north-star target of BASELINE.json on ONE MI355X, recorded: `psmc -N25 -t15 -r5 -p "4+25*2+4+6"` on the 30 M-bin
genome (README:12 of the reference) PLUS 100 bootstrap replicates at -N25 (README:57-62, main.c:16-20, aux.c:8-47), exact
and fast mode: wall clock, per-iteration E / M split, deviation of the RS / TR lines of the fast mode from the exact run
(which is byte-identical to the reference binary at fixture and chromosome size), and what that extrapolates to on 8
GPUs -- stated as an extrapolation.  Also dumps the parameter trajectory of `psmc -p 64*2` for bench.py's n128 extra.

    python scripts/northstar.py gpurun_out/r03_northstar.json [tests/golden/traj_n128.json]
Environment: NS_REPLICATES (100), NS_ITERS (25), NS_SKIP_BOOT_EXACT=1 (fast bootstrap only).
"""
import json
import os
import re
import subprocess
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HOST = os.path.join(ROOT, "psmc_amd", "host")
import em_parity  # noqa: E402


def write_psmcfa(path, segs, prefix):
    conv = np.frombuffer(b"TKN", dtype=np.uint8)
    with open(path, "wb") as fh:
        for i, s in enumerate(segs):
            fh.write((">%s%d\n" % (prefix, i)).encode())
            t = conv[s]
            n60 = len(t) // 60 * 60
            fh.write(np.concatenate([t[:n60].reshape(-1, 60), np.full((n60 // 60, 1), 10, np.uint8)], axis=1).tobytes())
            if n60 < len(t):
                fh.write(t[n60:].tobytes() + b"\n")


def final_round(text):
    r = em_parity.parse_psmc(text)[-1]
    return r


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_northstar.json")
    traj128 = sys.argv[2] if len(sys.argv) > 2 else None
    n_rep = int(os.environ.get("NS_REPLICATES", "100")); iters = int(os.environ.get("NS_ITERS", "25"))
    from psmc_amd import sim
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
    a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
    tmp = os.environ.get("TMPDIR", "/tmp")
    res = {"target": "BASELINE.json north_star: one -N25 whole-genome run (n = 64, ~30 M bins) plus 100 bootstraps", "device": "1 x MI355X"}
    # ---- the main run: 90 segments, exact / fast / fast with full counts / fast E-step + the reference's objective
    lens = sim.human_like_lengths(30_000_000, n_seg=90)
    segs = sim.simulate_genome(a, e, a0, lens, seed=43)
    gpath = os.path.join(tmp, "genome30m.psmcfa")
    write_psmcfa(gpath, segs, "seg")
    em_parity.ARGS[:] = ["-N%d" % iters, "-t15", "-r5", "-p", "4+25*2+4+6"]
    main_run = em_parity.compare("synthetic genome, 90 segments, %d bins (bench.py's workload)" % int(lens.sum()), gpath, tmp)
    main_run.pop("exact_trajectory", None)
    res["main_run"] = main_run
    json.dump(res, open(out_json, "w"), indent=1)
    # ---- 100 bootstrap replicates over splitfa trunks, all at -N25
    lens22 = sim.human_like_lengths(30_000_000, n_seg=22)
    chroms = sim.simulate_genome(a, e, a0, lens22, seed=43)
    trunks = []
    for s in chroms:            # utils/splitfa.c:20-35: 500 k-bin trunks, a tail shorter than 1.5 trunks stays whole
        L, pos = len(s), 0
        while L - pos >= 750_000:
            trunks.append(s[pos:pos + 500_000]); pos += 500_000
        trunks.append(s[pos:])
    spath = os.path.join(tmp, "split.psmcfa")
    write_psmcfa(spath, trunks, "t")
    args = ["-N%d" % iters, "-t15", "-r5", "-p", "4+25*2+4+6", spath]
    boot = {"workload": "%d trunks (%d bins, longest %d), %d replicates, %s" % (len(trunks), sum(len(t) for t in trunks), max(len(t) for t in trunks), n_rep, " ".join(args[:-1])), "runs": {}}
    modes = ("fast",) if os.environ.get("NS_SKIP_BOOT_EXACT") else ("exact", "fast")
    for mode in modes:
        env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1")
        t0 = time.time()
        r = subprocess.run([os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, "ns_%s-%%d.psmc" % mode), "--"] + args,
                           capture_output=True, text=True, env=env)
        wall = time.time() - t0
        its = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms", r.stderr)]
        es = np.array([x for x, _ in its]); ms = np.array([y for _, y in its])
        boot["runs"][mode] = dict(rc=r.returncode, wall_s=round(wall, 2), iterations=len(its), esteps_ms_first=float(es[0]) if len(es) else None,
                                  esteps_ms_median_later=float(np.median(es[2:])) if len(es) > 2 else None, msteps_ms_median=float(np.median(ms)) if len(ms) else None,
                                  per_iteration_ms=[dict(esteps=x, msteps=y) for x, y in its], stderr_tail=r.stderr[-400:] if r.returncode else "")
        sys.stderr.write("[northstar] psmc_boot %s: %.1f s\n" % (mode, wall))
        json.dump(dict(res, bootstrap=boot), open(out_json, "w"), indent=1)
    if len(modes) == 2:   # the same seeds draw the same resamples: replicate r of the fast run against replicate r of the exact run
        dev_lam, dev_tr, dev_lk = [], [], []
        for r in range(n_rep):
            try:
                x = final_round(open(os.path.join(tmp, "ns_exact-%d.psmc" % r)).read()); f = final_round(open(os.path.join(tmp, "ns_fast-%d.psmc" % r)).read())
            except Exception:
                continue
            dev_lam.append(em_parity.rel(f["rs_lam"], x["rs_lam"])); dev_tr.append(max(em_parity.rel(f["theta"], x["theta"]), em_parity.rel(f["rho"], x["rho"])))
            dev_lk.append(em_parity.rel(f["LK"], x["LK"]))
        if dev_lam:
            boot["fast_vs_exact_final_round"] = dict(replicates=len(dev_lam), RS_lambda_rel_dev_max=float(max(dev_lam)), RS_lambda_rel_dev_median=float(np.median(dev_lam)),
                                                     TR_rel_dev_max=float(max(dev_tr)), LK_rel_dev_max=float(max(dev_lk)))
    res["bootstrap"] = boot
    # ---- what the pieces add up to, and the 8-GPU extrapolation (NOT measured: no 8-GPU node from the build session)
    try:
        mr = main_run["runs"]
        tot = {m: round(mr[k]["wall_s"] + boot["runs"][m]["wall_s"], 1) for m, k in (("exact", "exact"), ("fast", "fast")) if m in boot["runs"]}
        res["one_gpu_total_s"] = dict(tot, note="main run + 100 bootstraps, one after the other on one GPU; process start-up, input parsing and output included")
        res["eight_gpu_extrapolation"] = dict(
            note="EXTRAPOLATED, not measured: bootstrap replicates are independent EM runs, psmc_boot deals them round robin over the visible devices (no collective), so 100 "
                 "replicates on 8 GPUs are 13 per device; the main run's E-step shards by segment (bench.py shard_sweep: predicted_speedup at 8 GPUs) but its M-step (host) does not",
            bootstrap_s={m: round(boot["runs"][m]["wall_s"] * 13.0 / n_rep, 1) for m in boot["runs"]})
    except Exception as ex_:
        res["totals_error"] = str(ex_)
    json.dump(res, open(out_json, "w"), indent=1)
    # ---- config 5 trajectory for bench.py's n128 extra: PA lines of `psmc -N25 -p 64*2` (fast mode) on the same genome
    if traj128:
        env = dict(os.environ, PSMC_HIP_MODE="fast", PSMC_TIMING="1")
        t0 = time.time()
        r = subprocess.run([os.path.join(HOST, "psmc"), "-N%d" % iters, "-t15", "-r5", "-p", "64*2", gpath], cwd=tmp, capture_output=True, text=True, env=env)
        if r.returncode == 0:
            rounds = em_parity.parse_psmc(r.stdout)
            tim = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", r.stderr)]
            json.dump(dict(pattern="64*2", source="scripts/northstar.py: PSMC_HIP_MODE=fast psmc -N%d -t15 -r5 -p 64*2 on the 30 M-bin synthetic genome (seed 43)" % iters,
                           rounds=[dict(round=q["round"], params=[q["theta"], q["rho"], q["max_t"]] + q["lam"]) for q in rounds]), open(traj128, "w"), indent=0)
            res["n128_run"] = dict(wall_s=round(time.time() - t0, 2), estep_ms_median_rounds_2plus=float(np.median([t[0] for t in tim[1:]])) if len(tim) > 1 else None,
                                   mstep_ms_median=float(np.median([t[1] for t in tim])) if tim else None)
        else:
            res["n128_run"] = dict(error=r.stderr[-400:])
        json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("main_run", "bootstrap")}, indent=1))


if __name__ == "__main__":
    main()
