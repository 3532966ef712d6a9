"""Real-data-shaped input: N runs of 2e4 and 2e5 bins, a 5e4-bin run of homozygosity, a het-dense stretch, a
segment that is one long N run with 100 called bins at each end -- tests/golden/make_golden_stress.py plants them in a
2.2 M-bin, 6-segment input and dumps the REAL reference's statistics at the parameters of three consecutive EM rounds and its
`psmc -N3` output.  Exact mode: every bit.  Fast mode: inside its stated tolerance with the plan a genome gets (3712-bin tiles,
3072-bin warm-ups, two launches of the back half: the GENOME options) and with the default plan for this size, parameters
moving from round to round on ONE context (the learned runs of round 0 meet the parameters of rounds 1 and 2); what the
verify / repair net had to do is printed.  And: a fast run whose tile boundaries cannot converge falls back to the exact
kernels for that E-step instead of ending (khmm.c has no such failure mode)."""
import gzip
import os
import subprocess
import numpy as np
import pytest
from conftest import bits_equal, GOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "psmc_amd", "host")
STRESS = os.path.join(GOLD, "stress")
GENOME = dict(chunk=3712, two_phase=2, merge1=0, warm_shift=1, kc_sub=4)   # what plan_fast picks for a 30 M-bin genome


@pytest.fixture(scope="module")
def hip():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "psmc_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    from psmc_amd import hip as h
    assert h.load_library().psmc_hip_device_count() > 0, "GPU tests need a visible HIP device"
    return h


@pytest.fixture(scope="module")
def stress():
    lut = np.full(256, 2, np.uint8); lut[ord("T")] = 0; lut[ord("K")] = 1   # the generator writes T / K / N only (cli.c:15-32)
    segs, cur = [], []
    for line in gzip.open(os.path.join(STRESS, "stress.psmcfa.gz"), "rb"):
        if line.startswith(b">"):
            if cur: segs.append(np.concatenate(cur))
            cur = []
        else:
            cur.append(lut[np.frombuffer(line.rstrip(b"\n"), dtype=np.uint8)])
    segs.append(np.concatenate(cur))
    g = dict(np.load(os.path.join(STRESS, "stress_estep.npz")))
    assert [len(s) for s in segs] == [700000, 500000, 400000, 250000, 150000, 200200]
    assert int((segs[0] == 2)[150000:350000].sum()) == 200000 and int((segs[5] == 2).sum()) == 200000
    return segs, g


def test_stress_exact_bit_identical(hip, stress):
    segs, g = stress
    es = hip.HipEStep(64, mode=hip.MODE_EXACT)
    es.load_segments(segs)
    for rd in range(3):
        k = "rd%d" % rd
        r = es.estep(g[k + ".a"], g[k + ".e"], g[k + ".a0"])
        assert bits_equal(r["A"], g[k + ".A"]) and bits_equal(r["E"], g[k + ".E"]) and r["LL"] == float(g[k + ".LL"]), rd
        assert bits_equal(r["chk"], g[k + ".seg_chk"])
    es.close()


def test_stress_psmc_binary_byte_identical(hip):
    args = open(os.path.join(STRESS, "stress_N3.args")).read().split()
    r = subprocess.run([os.path.join(HOST, "psmc")] + args, cwd=STRESS, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == gzip.open(os.path.join(STRESS, "stress_N3.psmc.gz"), "rt").read()


@pytest.mark.parametrize("plan", ["genome", "default", "genome_factored", "genome_gap_tiles_off", "chunk1001", "chunk1001_factored"])
def test_stress_fast_within_tolerance(hip, stress, plan):
    """... and the repairs STOP (round 5, "gap_tiles"): inside a run of missing data the chain forgets at the rate of the matrix's second
    eigenvalue alone (~5e5 bins), so every tile of a gap hangs on the vector that entered it; until round 4 "group_cap" cut the 2e5-bin gaps
    into two runs and the second one's head mis-speculated and cascaded through the rest in EVERY E-step (19 + 35 repair rounds, 64 ms
    instead of 9: profiles/r05_stress_timing.json).  With the gap tiles glued at plan time the plan has learned the input by the third
    E-step; gap_tiles=0 keeps the old behaviour (correct, slow)."""
    from test_gpu_estep import check_fast, relmax, tri_sums, FAST_TOL_STATS, FAST_TOL_LL
    segs, g = stress
    opts = dict(GENOME) if plan.startswith("genome") else {}
    if plan.endswith("gap_tiles_off"): opts["gap_tiles"] = 0
    # a tile length that is not a multiple of four: the gap tiles of a direction then do NOT cover the same steps (the backward
    # matrix of a tile starts at its first normalising position), so they cannot share one transfer matrix
    if plan.startswith("chunk1001"): opts = dict(chunk=1001, two_phase=2, merge1=0, warm_shift=1)
    es = hip.HipEStep(64, mode=hip.MODE_FAST, **opts)
    es.load_segments(segs)
    log = []
    for rd in (0, 1, 2, 0):   # ... and back to round 0's parameters on the plan the others shaped
        k = "rd%d" % rd
        p = dict(a=g[k + ".a"], e=g[k + ".e"], a0=g[k + ".a0"])
        o = dict(A=g[k + ".A"], E=g[k + ".E"], LL=float(g[k + ".LL"]))
        if plan.endswith("factored"):
            f = es.estep_factored(p["a"], p["e"], p["a0"])
            assert relmax(f["sums"], tri_sums(o["A"])) < FAST_TOL_STATS and relmax(f["E"], o["E"]) < FAST_TOL_STATS
            assert abs(f["LL"] - o["LL"]) <= FAST_TOL_LL * abs(o["LL"])
        else:
            check_fast(es.estep(p["a"], p["e"], p["a0"]), o, p)
        d = es.fast_diag(); pl = es.fast_plan()
        assert d["structured"] and d["warm_err_fwd"] <= 1e-12 and d["warm_err_bwd"] <= 1e-12, d
        log.append("rd%d: tiles %d x %d, repair rounds %d+%d (tiles %d+%d), glued %d/%d, longest warm-up %d/%d" % (
            rd, pl["tiles"], pl["tile_len"], d["fwd_rounds"], d["bwd_rounds"], d["fwd_tiles"], d["bwd_tiles"], pl["glued_fwd"], pl["glued_bwd"],
            pl["warm_fwd_max"], pl["warm_bwd_max"]))
    if plan.endswith("gap_tiles_off"): assert d["fwd_rounds"] + d["bwd_rounds"] >= 10, d    # what rounds 1-4 did on every E-step
    elif plan.startswith("chunk1001"): assert d["fwd_rounds"] + d["bwd_rounds"] <= 4, d       # (2200 tiles of 1001 bins under 3072-bin warm-ups: still learning)
    else: assert d["fwd_rounds"] + d["bwd_rounds"] <= 1, d                                   # the plan has learned the input
    print("\nstress, %s plan:\n  " % plan + "\n  ".join(log))
    rec = os.path.join(ROOT, "gpurun_out")   # what the verify / repair net had to do, kept when the suite runs on the GPU box (-> profiles/r05_stress_fast.txt)
    if os.path.isdir(rec):
        with open(os.path.join(rec, "stress_fast_%s.txt" % plan), "w") as fh:
            fh.write("stress fixture (tests/golden/stress), fast mode, %s plan, parameters of EM rounds 0, 1, 2, 0 on one context:\n  " % plan + "\n  ".join(log) + "\n")
    es.close()


def test_fast_run_that_cannot_converge_falls_back_to_exact(hip):
    """`psmc` in fast mode with a plan that cannot converge (64-bin warm-ups, no repair round allowed): psmc_hip_estep_factored
    returns PSMC_HIP_ECONVERGE; the binary repeats that E-step with the exact kernels, says so on stderr, and finishes -- LK of
    round 1 as close to the reference as any fast run's (the statistics of round 0 are then the exact ones)."""
    cli = os.path.join(GOLD, "cli")
    args = open(os.path.join(cli, "mid_n64_N4.args")).read().split()
    env = dict(os.environ, PSMC_HIP_MODE="fast", PSMC_HIP_OPTIONS="warmup=64,chunk=512,max_rounds=0,learn=0")
    r = subprocess.run([os.path.join(HOST, "psmc")] + args, cwd=cli, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-600:]
    assert r.stderr.count("repeating this E-step with the exact kernels") == 4, r.stderr[-600:]
    got = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("LK")]
    want = [float(l.split()[1]) for l in open(os.path.join(cli, "mid_n64_N4.psmc")).read().splitlines() if l.startswith("LK")]
    assert len(got) == len(want) == 5
    assert abs(got[1] - want[1]) <= 1e-7 * abs(want[1]) + 1e-6
    for x, y in zip(got[2:], want[2:]):
        assert abs(x - y) <= 1e-5 * abs(y)
    # the library itself still reports the failure (the fallback is the host driver's decision)
    from conftest import Golden
    gd = Golden(); p = gd.params("n64_curve")
    es = hip.HipEStep(64, mode=hip.MODE_FAST, warmup=64, chunk=512, max_rounds=0, learn=0)
    es.load_segments(gd.segs_mid)
    with pytest.raises(hip.HipError, match="converge"):
        es.estep(p["a"], p["e"], p["a0"])
    es.close()


def test_fast_bootstrap_that_cannot_converge_falls_back_per_replicate(hip, tmp_path):
    """The same through psmc_boot: a replicate whose fast E-step returns PSMC_HIP_ECONVERGE no longer ends the batch --
    psmc_hip_estep_batch repeats THAT replicate's E-step on an exact twin context over the same observations, says so once per replicate
    and EM iteration, and the job finishes; LK of every round as close to the exact-mode replicates as any fast run's."""
    def rounds(txt):
        return [float(l.split()[1]) for l in txt.splitlines() if l.startswith("LK")]
    args = ["-N2", "-t15", "-r5", "-p", "4+25*2+4+6", os.path.join(GOLD, "cli", "mid.psmcfa.gz")]
    boot = os.path.join(HOST, "psmc_boot")
    lk = {}
    for mode, opt in (("exact", ""), ("fast", "warmup=64,chunk=512,max_rounds=0,learn=0")):
        r = subprocess.run([boot, "-R", "3", "-S", "5", "-O", str(tmp_path / (mode + "-%d.psmc")), "--"] + args, capture_output=True, text=True,
                           env=dict(os.environ, PSMC_HIP_MODE=mode, PSMC_HIP_OPTIONS=opt))
        assert r.returncode == 0, r.stderr[-800:]
        if mode == "fast":
            assert r.stderr.count("repeating this E-step with the exact kernels") == 3 * 2, r.stderr[-800:]   # three replicates x two EM iterations
        lk[mode] = [rounds(open(tmp_path / ("%s-%d.psmc" % (mode, k))).read()) for k in range(3)]
    for ex, fa in zip(lk["exact"], lk["fast"]):
        assert len(ex) == len(fa) == 3
        for x, y in zip(ex[1:], fa[1:]):
            assert abs(x - y) <= 1e-6 * abs(x)
    # the library entry point itself: the replicate's statistics are the exact ones, the others stay fast
    from conftest import Golden
    from test_gpu_estep import check_fast
    import orc
    gd = Golden(); p = gd.params("n64_curve")
    es = hip.HipEStep(64, mode=hip.MODE_FAST, warmup=64, chunk=512, max_rounds=0, learn=0)
    es.load_segments(gd.segs_mid)
    sels = [[5, 4, 5, 3, 5], [0, 1, 2]]
    got = es.estep_batch([(p["a"], p["e"], p["a0"])] * 2, sels)
    es.close()
    ex = hip.HipEStep(64, mode=hip.MODE_EXACT)
    ex.load_segments(gd.segs_mid)
    for r_, sel in enumerate(sels):
        ex.select(sel)
        w = ex.estep(p["a"], p["e"], p["a0"])
        assert bits_equal(got["A"][r_], w["A"]) and bits_equal(got["E"][r_], w["E"]) and got["LL"][r_] == w["LL"], r_
    ex.close()
