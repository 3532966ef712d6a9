#!/usr/bin/env python3
"""End-to-end EM parity of the `psmc` drop-in on the MI355X box: -N25 -t15 -r5 -p "4+25*2+4+6" (README:12 of the
reference) on config 2 (tests/golden/full/chr22like.psmcfa.gz, 1 x 500 k bins, with the REAL reference's output as
golden) and on config 3 (the synthetic 30 M-bin genome bench.py times), in four configurations of the binary:

  exact            PSMC_HIP_MODE=exact                       bit-identical E-step, reference M-step
  fast             PSMC_HIP_MODE=fast                        factored statistics + O(N) objective (the fast default)
  fast_fullA       PSMC_HIP_MODE=fast PSMC_FACTORED=0        full counts on the matrix cores + O(N) objective
  fast_exactM      PSMC_HIP_MODE=fast PSMC_FAST_MSTEP=0      fast E-step, the reference's N*N-logarithm objective

and reports, per EM round, the largest relative deviation from the exact run of LK, theta_0, rho_0 and the lambda_k
(from the PA line, 9 decimals), plus wall times per round (PSMC_TIMING).  Writes JSON to argv[1]
(default gpurun_out/r02_em_parity.json) and the exact run's parameter trajectory to argv[2] if given.
"""
import gzip
import json
import os
import re
import subprocess
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PSMC = os.path.join(ROOT, "psmc_amd", "host", "psmc")
FULL = os.path.join(ROOT, "tests", "golden", "full")
ARGS = ["-N25", "-t15", "-r5", "-p", "4+25*2+4+6"]

CONFIGS = {
    "exact": dict(PSMC_HIP_MODE="exact"),
    "fast": dict(PSMC_HIP_MODE="fast"),
    "fast_fullA": dict(PSMC_HIP_MODE="fast", PSMC_FACTORED="0"),
    "fast_exactM": dict(PSMC_HIP_MODE="fast", PSMC_FAST_MSTEP="0"),
}


def parse_psmc(text):
    """-> list of rounds: dict(LK, theta, rho, max_t, lam (free lambdas), rs_lam (per state), rs_t, IT)."""
    rounds, cur, it = [], None, None
    for line in text.splitlines():
        f = line.split("\t")
        if f[0] == "IT":
            it = int(f[1])
        elif f[0] == "RD":
            cur = dict(round=int(f[1]), rs_lam=[], rs_t=[], IT=it)
            rounds.append(cur)
        elif cur is None:
            continue
        elif f[0] == "LK":
            cur["LK"] = float(f[1])
        elif f[0] == "RS":
            cur["rs_t"].append(float(f[2])); cur["rs_lam"].append(float(f[3]))
        elif f[0] == "PA":
            v = f[1].split()
            p = [float(x) for x in v[1:]]
            cur["theta"], cur["rho"], cur["max_t"], cur["lam"] = p[0], p[1], p[2], p[3:]
    return rounds


def rel(x, y):
    x, y = np.asarray(x, float), np.asarray(y, float)
    return float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300)))


def run(cfg, infile, cwd):
    env = dict(os.environ, PSMC_TIMING="1", **CONFIGS[cfg])
    t0 = time.time()
    r = subprocess.run([PSMC] + ARGS + [infile], cwd=cwd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (cfg, r.stderr[-2000:]))
    tim = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", r.stderr)]
    return r.stdout, tim, wall


def compare(name, infile, cwd, ref_text=None):
    res = dict(input=name, args=" ".join(ARGS), runs={})
    outs = {}
    for cfg in CONFIGS:
        out, tim, wall = run(cfg, infile, cwd)
        outs[cfg] = parse_psmc(out)
        e = np.array([t[0] for t in tim]); m = np.array([t[1] for t in tim])
        res["runs"][cfg] = dict(wall_s=round(wall, 2), estep_ms_first=float(e[0]), estep_ms_median_rounds_2plus=float(np.median(e[1:])),
                                estep_ms_max_rounds_2plus=float(e[1:].max()), mstep_ms_median=float(np.median(m)),
                                em_iteration_ms_median_rounds_2plus=float(np.median(e[1:] + m[1:])),
                                IT=[r["IT"] for r in outs[cfg][1:]], LK_final=outs[cfg][-1]["LK"])
        if cfg == "exact" and ref_text is not None:
            res["exact_vs_reference_binary"] = "byte-identical" if out == ref_text else "DIFFERENT"
        sys.stderr.write("[em_parity] %s %s: %.1f s\n" % (name, cfg, wall))
    ex = outs["exact"]
    for cfg in CONFIGS:
        if cfg == "exact":
            continue
        per = []
        for a, b in zip(outs[cfg], ex):
            per.append(dict(round=a["round"], LK=rel(a["LK"], b["LK"]), theta=rel(a["theta"], b["theta"]), rho=rel(a["rho"], b["rho"]),
                            lam_max=rel(a["lam"], b["lam"]), rs_lam_max=(rel(a["rs_lam"], b["rs_lam"]) if a["rs_lam"] and len(a["rs_lam"]) == len(b["rs_lam"]) else None),
                            rs_t_max=(rel(a["rs_t"][1:], b["rs_t"][1:]) if len(a["rs_t"]) > 1 and len(a["rs_t"]) == len(b["rs_t"]) else None), lam_median=float(np.median(np.abs(np.array(a["lam"]) - np.array(b["lam"])) / np.array(b["lam"])))))
        res["runs"][cfg]["deviation_vs_exact_per_round"] = per
        res["runs"][cfg]["max_over_rounds"] = {k: max(p[k] for p in per) for k in ("LK", "theta", "rho", "lam_max")}
        res["runs"][cfg]["final_round"] = per[-1]
        # which lambda deviates most in the final round, and how well the data determine it (expected coalescences there)
        la, lb = np.array(outs[cfg][-1]["lam"]), np.array(ex[-1]["lam"])
        res["runs"][cfg]["final_lambda_rel_dev"] = [float(x) for x in np.abs(la - lb) / lb]
    res["exact_trajectory"] = [dict(round=r["round"], params=[r["theta"], r["rho"], r["max_t"]] + r["lam"]) for r in ex]
    return res


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_em_parity.json")
    traj_json = sys.argv[2] if len(sys.argv) > 2 else None
    which = os.environ.get("EM_PARITY_INPUTS", "chr22like,genome").split(",")
    os.makedirs(os.path.dirname(out_json), exist_ok=True)
    results = {}
    if "chr22like" in which:
        ref = gzip.open(os.path.join(FULL, "chr22like_N25.psmc.gz"), "rt").read()
        results["config2_chr22like_500k"] = compare("tests/golden/full/chr22like.psmcfa.gz (1 x 500,000 bins)", "chr22like.psmcfa.gz", FULL, ref)
    if "genome" in which:
        from psmc_amd import sim
        g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
        a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
        lens = sim.human_like_lengths(30_000_000, n_seg=90)
        segs = sim.simulate_genome(a, e, a0, lens, seed=43)
        tmp = os.environ.get("TMPDIR", "/tmp")
        path = os.path.join(tmp, "genome30m.psmcfa")
        conv = np.frombuffer(b"TKN", dtype=np.uint8)
        with open(path, "wb") as fh:
            for i, s in enumerate(segs):
                fh.write((">seg%d\n" % i).encode())
                t = conv[s]
                n60 = len(t) // 60 * 60
                body = np.concatenate([t[:n60].reshape(-1, 60), np.full((n60 // 60, 1), 10, np.uint8)], axis=1).tobytes()
                fh.write(body)
                if n60 < len(t):
                    fh.write(t[n60:].tobytes() + b"\n")
        results["config3_genome_30m"] = compare("synthetic genome, 90 segments, %d bins (bench.py's workload)" % int(lens.sum()), path, tmp)
        traj = results["config3_genome_30m"]["exact_trajectory"]
        if traj_json:
            json.dump(dict(pattern="4+25*2+4+6", source="scripts/em_parity.py: PSMC_HIP_MODE=exact psmc -N25 -t15 -r5 on the 30 M-bin synthetic genome (seed 43)",
                           rounds=traj), open(traj_json, "w"), indent=0)
    json.dump(results, open(out_json, "w"), indent=1)
    for k, v in results.items():
        for cfg, r in v["runs"].items():
            if "max_over_rounds" in r:
                sys.stderr.write("%s %s max over rounds: %s\n" % (k, cfg, r["max_over_rounds"]))


if __name__ == "__main__":
    main()
