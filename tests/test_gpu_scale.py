"""Parity at the sizes BASELINE.json quotes (no oracle finishes there: the reference needs ~2 minutes per E-step of a
30 M-bin genome): size-independent properties of the sufficient statistics, cross-checks between the independent code
paths of the library (full counts vs factored sums, fast vs exact on the three longest segments -- the exact mode is
pinned bit for bit to the reference at fixture size by tests/test_gpu_estep.py), on the very workload bench.py times
(config 3: 90 segments, 30 M bins, 64 states) and on config 5 (128 states, whole genome)."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_STATS = 1e-10   # as tests/test_gpu_estep.py: max |x - ref| / max |ref|
TOL_LL = 1e-12
TOL_SUM = 1e-9      # sum A = sum (L - 1) etc.: 3e7 terms


@pytest.fixture(scope="module")
def hip():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "psmc_amd", "csrc")], check=True)
    from psmc_amd import hip as h
    assert h.load_library().psmc_hip_device_count() > 0, "GPU tests need a visible HIP device"
    return h


@pytest.fixture(scope="module")
def genome(golden):
    """bench.py's rank-0 workload: same lengths, same seed."""
    from psmc_amd import sim
    p = golden.params("n64_curve")
    lens = sim.human_like_lengths(30_000_000, n_seg=90)
    assert lens.max() == 2_490_000
    return sim.simulate_genome(p["a"], p["e"], p["a0"], lens, seed=43)


def relmax(x, y):
    return float(np.abs(np.asarray(x) - np.asarray(y)).max() / np.abs(np.asarray(y)).max())


def tri_sums(A):
    lo, up = np.tril(A, -1), np.triu(A, 1)
    return np.stack([lo.sum(1), up.sum(1), np.diag(A).copy(), lo.sum(0), up.sum(0)])


def check_totals(r, segs, sel=None):
    ss = segs if sel is None else [segs[i] for i in sel]
    tot = float(sum(len(s) - 1 for s in ss))
    nonmiss = float(sum(int((s[:-1] != 2).sum()) for s in ss))
    het = float(sum(int((s[:-1] == 1).sum()) for s in ss))
    assert abs(r["A"].sum() - tot) < TOL_SUM * tot, (r["A"].sum(), tot)        # every transition 1..L-1 counted once
    assert abs(r["E"].sum() - nonmiss) < TOL_SUM * nonmiss, (r["E"].sum(), nonmiss)  # khmm.c:355: the missing row is dropped
    assert abs(r["E"][1].sum() - het) < TOL_SUM * max(het, 1.0)                # posterior mass at het positions
    assert (r["A"] > 0).all() and (r["E"] > 0).all() and np.isfinite(r["LL"]) and r["LL"] < 0


def test_config3_full_size_n64(hip, golden, genome):
    """30,000,001 bins x 64 states in 90 segments -- the step bench.py times, with moving parameters."""
    segs = genome
    fa = hip.HipEStep(64, mode=hip.MODE_FAST)
    fa.load_segments(segs)
    ex = hip.HipEStep(64, mode=hip.MODE_EXACT)
    longest = [0, 1, 2]
    ex.load_segments([segs[i] for i in longest])
    for key in ("n64_curve", "n64_flat", "n64_curve"):   # the plan learned for one parameter set meets another
        p = golden.params(key)
        r = fa.estep(p["a"], p["e"], p["a0"])
        d = fa.fast_diag()
        assert d["structured"] and d["back_half"] == 1
        assert d["warm_err_fwd"] <= 1e-12 and d["warm_err_bwd"] <= 1e-12, d
        check_totals(r, segs)
        # the factored back half is a different kernel family (no counts GEMM): same triangular sums, E, LL
        f = fa.estep_factored(p["a"], p["e"], p["a0"])
        assert relmax(f["sums"], tri_sums(r["A"])) < TOL_STATS and relmax(f["E"], r["E"]) < TOL_STATS
        assert abs(f["LL"] - r["LL"]) <= TOL_LL * abs(r["LL"])
        # additivity over a partition of the segments (hmm_add_expect is a plain sum, khmm.c:346-359)
        half, other = list(range(0, 90, 2)), list(range(1, 90, 2))
        fa.select(half); ra = fa.estep(p["a"], p["e"], p["a0"])
        fa.select(other); rb = fa.estep(p["a"], p["e"], p["a0"])
        assert relmax(ra["A"] + rb["A"], r["A"]) < TOL_STATS and relmax(ra["E"] + rb["E"], r["E"]) < TOL_STATS
        assert abs(ra["LL"] + rb["LL"] - r["LL"]) < 1e-11 * abs(r["LL"])
        # fast vs exact (bit-identical to khmm.c) on the three longest segments: 2.49e6, 2.43e6, 1.98e6 bins
        fa.select(longest); rf = fa.estep(p["a"], p["e"], p["a0"])
        rx = ex.estep(p["a"], p["e"], p["a0"])
        check_totals(rx, segs, longest)
        assert relmax(rf["A"], rx["A"]) < TOL_STATS and relmax(rf["E"], rx["E"]) < TOL_STATS
        assert abs(rf["LL"] - rx["LL"]) <= TOL_LL * abs(rx["LL"])
        assert np.abs(rx["chk"] - 1.0).max() < 1e-6   # khmm.c:237-240: no underflow warning at 2.49e6 bins
        fa.select(list(range(90)))
    fa.close(); ex.close()


def test_config5_full_size_n128(hip, golden, genome):
    """-p "64*2": 30 M bins x 128 states.  Properties, additivity, fast vs exact on a long, a medium and a short segment."""
    g, k = golden.n128, "n128_curve"
    a, e, a0 = g[k + ".a"], g[k + ".e"], g[k + ".a0"]
    segs = genome
    fa = hip.HipEStep(128, mode=hip.MODE_FAST)
    fa.load_segments(segs)
    r = fa.estep(a, e, a0)
    assert fa.fast_diag()["structured"]
    check_totals(r, segs)
    r2 = fa.estep(a, e, a0)   # learned plan, same parameters
    assert relmax(r2["A"], r["A"]) < TOL_STATS and abs(r2["LL"] - r["LL"]) <= TOL_LL * abs(r["LL"])
    f = fa.estep_factored(a, e, a0)   # the O(N) statistics (what the psmc binary uses in fast mode), eight states per lane
    assert relmax(f["sums"], tri_sums(r["A"])) < TOL_STATS and relmax(f["E"], r["E"]) < TOL_STATS
    assert abs(f["LL"] - r["LL"]) <= TOL_LL * abs(r["LL"])
    half, other = list(range(0, 90, 2)), list(range(1, 90, 2))
    fa.select(half); ra = fa.estep(a, e, a0)
    fa.select(other); rb = fa.estep(a, e, a0)
    assert relmax(ra["A"] + rb["A"], r["A"]) < TOL_STATS and abs(ra["LL"] + rb["LL"] - r["LL"]) < 1e-11 * abs(r["LL"])
    pick = [0, 12, 40]
    fa.select(pick); rf = fa.estep(a, e, a0)
    ex = hip.HipEStep(128, mode=hip.MODE_EXACT)
    ex.load_segments([segs[i] for i in pick])
    rx = ex.estep(a, e, a0)
    check_totals(rx, segs, pick)
    assert relmax(rf["A"], rx["A"]) < TOL_STATS and relmax(rf["E"], rx["E"]) < TOL_STATS
    assert abs(rf["LL"] - rx["LL"]) <= TOL_LL * abs(rx["LL"])
    fa.close(); ex.close()


def moving_params():
    """(a, e, a0) of consecutive EM rounds of `psmc -N25` on the benchmark genome (what bench.py cycles through)."""
    import json
    from psmc_amd import hostlib
    tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
    return [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"] if r["round"] >= 1]


@pytest.mark.parametrize("workload", ["config2_500k", "share_1of8"])
def test_shard_sized_inputs_one_round_plan(hip, golden, genome, workload):
    """the sizes a strong-scaling run and config 2 put on one GPU -- rank 0's LPT share of the genome at 8
    GPUs (3.75 M bins) and one 500 k-bin chromosome.  The planner gives them ONE round of the fused back half (every tile
    speculating, phase 1 as one grid, one launch of the counts).  Fourteen EM
    rounds of moving parameters: totals, factored vs full counts every round, fast vs exact (bit-identical to khmm.c)
    in rounds 0, 6 and 13 -- on everything (500 k) or on the two longest segments (share)."""
    from psmc_amd import sim
    from psmc_amd.dist import partition_segments
    p0 = golden.params("n64_curve")
    if workload == "config2_500k":
        segs = [sim.simulate_segment(p0["a"], p0["e"], p0["a0"], 500_000, np.random.default_rng(7))]
        pick = [0]
    else:
        lens = np.array([len(s) for s in genome])
        segs = [genome[i] for i in partition_segments(lens, 8)[0]]
        pick = list(np.argsort([-len(s) for s in segs])[:2])
    fa = hip.HipEStep(64, mode=hip.MODE_FAST)
    fa.load_segments(segs)
    fb = fa
    if len(pick) < len(segs):   # a context of its own for the segments the exact mode checks (a selection would re-plan and forget the learned warm-ups)
        fb = hip.HipEStep(64, mode=hip.MODE_FAST)
        fb.load_segments([segs[i] for i in pick])
    ex = hip.HipEStep(64, mode=hip.MODE_EXACT)
    ex.load_segments([segs[i] for i in pick])
    traj = moving_params()
    w0 = None
    for it in range(14):
        a, e, a0 = traj[it]
        r = fa.estep(a, e, a0)
        d = fa.fast_diag()
        assert d["structured"] and d["back_half"] == 1 and d["merged_phase1"] and d["fused_launches"] == 1, d
        assert d["n_chunks"] <= 4096
        assert d["warm_err_fwd"] <= 1e-12 and d["warm_err_bwd"] <= 1e-12, d
        check_totals(r, segs)
        f = fa.estep_factored(a, e, a0)
        assert relmax(f["sums"], tri_sums(r["A"])) < TOL_STATS and relmax(f["E"], r["E"]) < TOL_STATS
        assert abs(f["LL"] - r["LL"]) <= TOL_LL * abs(r["LL"])
        rf = r if fb is fa else fb.estep(a, e, a0)
        if it in (0, 6, 13):
            rx = ex.estep(a, e, a0)
            assert relmax(rf["A"], rx["A"]) < TOL_STATS and relmax(rf["E"], rx["E"]) < TOL_STATS, (it, relmax(rf["A"], rx["A"]))
            assert abs(rf["LL"] - rx["LL"]) <= TOL_LL * abs(rx["LL"])
        pl = fa.fast_plan()
        w0 = w0 or pl
    assert pl["tiles"] == d["n_chunks"] and pl["warm_fwd_max"] <= 3072 and pl["warm_bwd_max"] <= 3072, (w0, pl)   # one round: a failed tile is glued, never doubled
    if fb is not fa:
        fb.close()
    fa.close(); ex.close()


def test_bench_two_ranks_on_one_device():
    """The driver's N > 1 command line (torch.distributed.run, one rank per GPU) on a box with one GPU: BENCH_SINGLE_GPU_TEST=1 puts
    both ranks on device 0 (gloo instead of RCCL).  Checks the contract line, the strong-scaling extra and the child process that
    runs the C library's own multi-device engine (psmc_hip_group_*) beside it."""
    import json
    import sys
    env = dict(os.environ, BENCH_SINGLE_GPU_TEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--bins", "3000000", "--segments", "9"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]   # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["warmup"] == 2 and j["unit"] == "bins/s"
    cm = j["config"]["comm"]   # what a SCALE record needs to say by itself: the backend, how many ranks one all-reduce counted, where they sat
    assert cm["backend"] == "gloo" and cm["world_size"] == 2 and cm["ranks_in_allreduce"] == 2, cm
    assert [d["rank"] for d in cm["rank_devices"]] == [0, 1] and cm["distinct_devices"] == 1, cm   # (BENCH_SINGLE_GPU_TEST: both on device 0)
    from psmc_amd import sim
    genome_bins = int(sim.human_like_lengths(3_000_000, n_seg=9).sum())
    assert abs(j["value"] - 2 * genome_bins / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]   # whole-job aggregate: both ranks' genomes
    assert j["strong_scaling"]["bins_total"] == genome_bins and j["strong_scaling"]["value"] > 0
    for sc, bins in (("weak", 2 * genome_bins), ("strong", genome_bins)):
        g = j["group_engine"][sc]
        assert "error" not in g, g
        assert g["bins_total"] == bins and g["devices"] == [0, 0] and g["value"] > 0
        assert g["selfcheck"]["shards"] == 2 and g["selfcheck"]["path"] == "host_sum", g   # two shards on one device: no communicator
