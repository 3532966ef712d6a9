#!/usr/bin/env python3
"""The north-star target of BASELINE.json on ONE MI355X, recorded: `psmc -N25 -t15 -r5 -p "4+25*2+4+6"` on the 30 M-bin
genome (README:12 of the reference) PLUS 100 bootstrap replicates at -N25 on its splitfa trunks (README:49-62, main.c:16-20,
aux.c:8-47).  Since round 5 that is ONE job -- `psmc_boot --main`: the main run on a thread of its own beside the replicates'
batched E-steps -- in exact mode (every byte as the reference would write it) and in fast mode; beside it, on request, the
two programs one after the other as in round 4.  Reports wall clock, per-iteration E / M split, the main run's E-steps,
whether the main output of the joint job equals `psmc`'s own byte for byte, the deviation of the RS / TR lines of the fast
mode from the exact run, and what that extrapolates to on 8 GPUs -- stated as an extrapolation.

    python scripts/northstar.py gpurun_out/r05_northstar.json
Environment: NS_REPLICATES (100), NS_ITERS (25), NS_MODES ("fast,exact": the fast job first -- a job started right after the exact one waits ~3.5 s in its first device allocations while the driver wipes the 250 GB that job released, profiles/r06_fast_after_exact.txt), NS_SEPARATE=1 (also time `psmc` alone, exact and
fast, and check the joint job's main output against it), NS_MAIN_CUS (PSMC_BOOT_MAIN_CUS of the exact job; default: unset),
NS_TRAJ128=<path> (dump the parameter trajectory of `psmc -p 64*2` for bench.py's n128 extra).
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
import northstar_data as nd  # noqa: E402

ARGS = ["-t15", "-r5", "-p", "4+25*2+4+6"]


def final_round(text):
    return em_parity.parse_psmc(text)[-1]


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_northstar.json")
    n_rep = int(os.environ.get("NS_REPLICATES", "100")); iters = int(os.environ.get("NS_ITERS", "25"))
    modes = os.environ.get("NS_MODES", "fast,exact").split(",")
    tmp = os.environ.get("TMPDIR", "/tmp")
    f = nd.files(tmp)
    res = {"target": "BASELINE.json north_star: one -N25 whole-genome run (n = 64, ~30 M bins) plus 100 bootstraps", "device": "1 x MI355X",
           "workload": "main run: 90 segments, 30,000,001 bins, longest 2,490,000; replicates: %d trunks (%d bins, longest %d), %d replicates; -N%d %s"
                       % (f["n_trunks"], f["trunk_bins"], f["longest_trunk"], n_rep, iters, " ".join(ARGS)),
           "one_schedule": {}}

    def save():
        json.dump(res, open(out_json, "w"), indent=1)
    # ---- the joint job
    for mode in modes:
        env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_SEED="4242")
        if mode == "exact" and os.environ.get("NS_MAIN_CUS"):
            env["PSMC_BOOT_MAIN_CUS"] = os.environ["NS_MAIN_CUS"]
        cmd = [os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, "ns_%s-%%d.psmc" % mode),
               "--main", os.path.join(tmp, "ns_%s-main.psmc" % mode), "--main-input", f["genome"], "--", "-N%d" % iters] + ARGS + [f["split"]]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        wall = time.time() - t0
        its = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms", r.stderr)]
        mes = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"\[psmc\] E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", r.stderr)]
        mm = re.search(r"main run: \d+ EM iterations beside the replicates in ([0-9.]+) ms", r.stderr)
        es = np.array([x for x, _ in its]); ms = np.array([y for _, y in its])
        res["one_schedule"][mode] = dict(
            rc=r.returncode, wall_s=round(wall, 2), iterations=len(its), esteps_ms_first=float(es[0]) if len(es) else None,
            esteps_ms_median_later=float(np.median(es[2:])) if len(es) > 2 else None, msteps_ms_median=float(np.median(ms)) if len(ms) else None,
            main_run=dict(total_s=float(mm.group(1)) / 1e3 if mm else None, estep_ms_first=mes[0][0] if mes else None, estep_ms_median=float(np.median([x for x, _ in mes[1:]])) if len(mes) > 1 else None,
                          mstep_ms_median=float(np.median([y for _, y in mes])) if mes else None),
            per_iteration_ms=[dict(esteps=x, msteps=y) for x, y in its], stderr_tail=r.stderr[-500:] if r.returncode else "")
        sys.stderr.write("[northstar] one schedule, %s: %.1f s\n" % (mode, wall))
        save()
    # ---- fast against exact, replicate by replicate (the same seeds draw the same resamples) and for the main run
    if "exact" in modes and "fast" in modes:
        dev_lam, dev_tr, dev_lk = [], [], []
        for r in list(range(n_rep)) + ["main"]:
            try:
                x = final_round(open(os.path.join(tmp, "ns_exact-%s.psmc" % r)).read()); fa = final_round(open(os.path.join(tmp, "ns_fast-%s.psmc" % r)).read())
            except Exception:
                continue
            d = (em_parity.rel(fa["rs_lam"], x["rs_lam"]), max(em_parity.rel(fa["theta"], x["theta"]), em_parity.rel(fa["rho"], x["rho"])), em_parity.rel(fa["LK"], x["LK"]))
            if r == "main":
                res["fast_vs_exact_main_final_round"] = dict(RS_lambda_rel_dev=d[0], TR_rel_dev=d[1], LK_rel_dev=d[2])
            else:
                dev_lam.append(d[0]); dev_tr.append(d[1]); dev_lk.append(d[2])
        if dev_lam:
            res["fast_vs_exact_final_round"] = dict(replicates=len(dev_lam), RS_lambda_rel_dev_max=float(max(dev_lam)), RS_lambda_rel_dev_median=float(np.median(dev_lam)),
                                                    TR_rel_dev_max=float(max(dev_tr)), LK_rel_dev_max=float(max(dev_lk)))
        save()
    # ---- the two programs one after the other (round 4's way), and the identity of the joint job's main output
    if os.environ.get("NS_SEPARATE"):
        sep = {}
        for mode in modes:
            env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_SEED="4242")
            t0 = time.time()
            r = subprocess.run([os.path.join(HOST, "psmc"), "-N%d" % iters] + ARGS + [f["genome"]], capture_output=True, text=True, env=env)
            w = time.time() - t0
            tim = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", r.stderr)]
            same = None
            try:
                same = open(os.path.join(tmp, "ns_%s-main.psmc" % mode)).read() == r.stdout
            except Exception:
                pass
            sep[mode] = dict(rc=r.returncode, psmc_wall_s=round(w, 2), estep_ms_median=float(np.median([x for x, _ in tim[1:]])) if len(tim) > 1 else None,
                             main_output_of_the_joint_job_is_byte_identical=same, estep_ms_first=tim[0][0] if tim else None)
            save()
        sep["note"] = ("psmc_wall_s of the FIRST program run here follows the exact joint job: its first allocations wait for the driver to clear the 250 GB that "
                       "job released (estep_ms_first holds the wait); on a quiet device the fast run takes 1.0 s in all (profiles/r06_fast_after_exact.txt)")
        res["separate_main_run"] = sep
    # ---- totals and the 8-GPU extrapolation (NOT measured: no 8-GPU node from a build session)
    try:
        os_ = res["one_schedule"]
        res["one_gpu_total_s"] = dict({m: os_[m]["wall_s"] for m in os_}, note="ONE job: main run + %d bootstraps, process start-up, input parsing, table allocation and output included" % n_rep,
                                      round4="exact 290 s (79.5 + 210.4, two programs one after the other), fast 34 s: profiles/r04b_northstar.json")
        per13 = {m: round(os_[m]["wall_s"] * 13.0 / n_rep, 1) for m in os_}
        main_alone = {m: (os_[m]["main_run"]["estep_ms_median"] or 0) * iters / 1e3 for m in os_}
        res["eight_gpu_extrapolation"] = dict(
            note="EXTRAPOLATED, not measured: replicates are independent EM runs, psmc_boot deals them round robin over the visible devices (no collective): 100 replicates on 8 GPUs "
                 "are 13 per device, and the main run rides on the first device as it does here -- the job then ends with max(main run, bootstraps / 8).  Exact mode does not shard "
                 "below its longest segment (one wave per segment), so the main run's %d x ~3.2 s is the floor" % iters,
            bootstraps_s=per13, main_run_alone_s={m: round(v, 1) for m, v in main_alone.items()},
            job_s={m: round(max(per13[m], main_alone[m]) + 10.0, 1) for m in os_})
    except Exception as ex_:
        res["totals_error"] = str(ex_)
    save()
    # ---- config 5 trajectory for bench.py's n128 extra: PA lines of `psmc -N25 -p 64*2` (fast mode) on the same genome
    traj128 = os.environ.get("NS_TRAJ128")
    if traj128:
        env = dict(os.environ, PSMC_HIP_MODE="fast", PSMC_TIMING="1")
        t0 = time.time()
        r = subprocess.run([os.path.join(HOST, "psmc"), "-N%d" % iters, "-t15", "-r5", "-p", "64*2", f["genome"]], cwd=tmp, capture_output=True, text=True, env=env)
        if r.returncode == 0:
            rounds = em_parity.parse_psmc(r.stdout)
            tim = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", r.stderr)]
            json.dump(dict(pattern="64*2", source="scripts/northstar.py: PSMC_HIP_MODE=fast psmc -N%d -t15 -r5 -p 64*2 on the 30 M-bin synthetic genome (seed 43)" % iters,
                           rounds=[dict(round=q["round"], params=[q["theta"], q["rho"], q["max_t"]] + q["lam"]) for q in rounds]), open(traj128, "w"), indent=0)
            res["n128_run"] = dict(wall_s=round(time.time() - t0, 2), estep_ms_median_rounds_2plus=float(np.median([t[0] for t in tim[1:]])) if len(tim) > 1 else None,
                                   mstep_ms_median=float(np.median([t[1] for t in tim])) if tim else None)
        else:
            res["n128_run"] = dict(error=r.stderr[-400:])
        save()
    print(json.dumps({k: v for k, v in res.items() if k != "one_schedule"}, indent=1))
    for m, v in res["one_schedule"].items():
        print(m, {k: v[k] for k in ("rc", "wall_s", "esteps_ms_first", "esteps_ms_median_later", "msteps_ms_median", "main_run")})


if __name__ == "__main__":
    main()
