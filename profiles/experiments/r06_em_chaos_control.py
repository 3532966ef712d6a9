"""Round 6, control for scripts/fuzz_em.py (CPU only, build container): on the inputs where fast mode's lambdas differ most from exact mode's,
how far do they move when the E-step is EXACT (the oracle) and only the rounding of the M-step's objective changes (PSMC_FAST_MSTEP=1: the O(N)
form of the same sum)?  Reference binary against the oracle-backed host driver (tests/test_host_cli.py builds it in /tmp/psmc_test_build)."""
import os, sys, json, subprocess, numpy as np
ROOT='/root/repo'; sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/scripts')
import northstar_data as nd, em_parity
from psmc_amd import hostlib, sim
tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
P = [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"][1:]]
for seed in (80, 31, 60):
    rng = np.random.default_rng(seed)
    p = P[int(rng.integers(len(P)))]
    total = int(np.exp(rng.uniform(np.log(100_000), np.log(1_500_000))))
    k = int(rng.integers(1, 11))
    w = rng.random(k) + 0.05; lens = np.maximum(200, (w / w.sum() * total).astype(int))
    segs = []
    for L in lens:
        s = sim.simulate_segment(p[0], p[1], p[2], int(L), rng)
        for _ in range(int(rng.integers(0, 3))):
            g = int(np.exp(rng.uniform(np.log(10), np.log(min(int(L) // 2, 40_000) + 11)))); at = int(rng.integers(0, int(L) - g + 1)); s[at:at + g] = 2
        segs.append(s)
    fa = "/tmp/em%d.psmcfa" % seed; nd.write_psmcfa(fa, segs, "s")
    args = ["-N10", "-t15", "-r5", "-p", "4+25*2+4+6", fa]
    a = subprocess.run([ROOT+"/oracle/_ref/psmc_ref"] + args, capture_output=True, text=True).stdout
    b = subprocess.run(["/tmp/psmc_test_build/psmc_oracle_backend"] + args, capture_output=True, text=True, env=dict(os.environ, PSMC_FAST_MSTEP="1")).stdout
    c = subprocess.run(["/tmp/psmc_test_build/psmc_oracle_backend"] + args, capture_output=True, text=True).stdout
    ra, rb, rc = em_parity.parse_psmc(a), em_parity.parse_psmc(b), em_parity.parse_psmc(c)
    print(seed, [int(x) for x in lens], "identical without the O(N) objective:", a == c)
    for i in (1, 5, 10):
        print("  round", i, "LK rel", abs(rb[i]["LK"]-ra[i]["LK"])/abs(ra[i]["LK"]), "lam rel", em_parity.rel(rb[i]["lam"], ra[i]["lam"]), "theta rel", abs(rb[i]["theta"]-ra[i]["theta"])/ra[i]["theta"])
