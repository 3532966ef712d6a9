"""Exact mode beyond 128 hidden states (129 .. 1024): the wide kernels of psmc_amd/csrc/estep_wide.hip against goldens of the
REAL reference at 200 (`-p "100*2"`) and 149 states (tests/golden/make_golden_wide.py) and against the oracle on random HMMs.
The reference has no limit (khmm.c:10-23, cli.c:66-99); VERDICT r4 "missing 1".  The end-to-end goldens
(tests/golden/cli/small_n200_N2, small_n149_d) run through the `psmc` binary in tests/test_host_cli.py."""
import os
import subprocess
import numpy as np
import pytest
from conftest import bits_equal, GOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "psmc_amd", "csrc")], check=True)
    from psmc_amd import hip as h
    assert h.load_library().psmc_hip_device_count() > 0, "GPU tests need a visible HIP device"
    return h


def random_hmm(rng, n):
    a = rng.random((n, n)) ** 4 * 0.02 + np.eye(n) * (0.9 + 0.1 * rng.random(n))
    a /= a.sum(1, keepdims=True)
    e = np.ones((3, n)); e[1] = 0.001 + rng.random(n) * 0.15; e[0] = 1.0 - e[1]
    a0 = rng.random(n) + 0.1; a0 /= a0.sum()
    return a, e, a0


@pytest.mark.parametrize("key", ["n200", "n149"])
def test_wide_golden(hip, golden, key):
    """Summed and per-segment statistics, log-likelihoods, underflow check values and the f / b / s tables of one segment:
    every double as the reference computes it at 200 and 149 states."""
    g = dict(np.load(os.path.join(GOLD, "estep_wide.npz")))
    a, e, a0 = g[key + ".a"], g[key + ".e"], g[key + ".a0"]
    n = a.shape[0]
    segs = golden.segs_small[:8]
    for mode in (hip.MODE_EXACT, hip.MODE_FAST):   # a fast-mode context runs the exact kernels beyond 128 states
        es = hip.HipEStep(n, mode=mode)
        es.load_segments(segs)
        r = es.estep(a, e, a0)
        assert bits_equal(r["A"], g[key + ".A"]) and bits_equal(r["E"], g[key + ".E"]) and bits_equal(r["A0"], g[key + ".A0"])
        assert r["LL"] == float(g[key + ".LL"])
        assert bits_equal(r["chk"], g[key + ".seg_chk"])
        s = es.estep_segments(a, e, a0)
        assert bits_equal(s["seg_E"], g[key + ".seg_E"]) and bits_equal(s["seg_LL"], g[key + ".seg_LL"])
        assert bits_equal(s["seg_A"].sum(2), g[key + ".seg_A_rowsum"]) and bits_equal(s["seg_A"].sum(1), g[key + ".seg_A_colsum"])
        f, b, sc = es.tables(5)
        assert bits_equal(f[::7], g[key + ".f65"]) and bits_equal(b[::7], g[key + ".b65"]) and bits_equal(sc, g[key + ".s65"])
        if mode == hip.MODE_FAST:
            with pytest.raises(hip.HipError):
                es.estep_factored(a, e[:2], a0)     # no factored / device-resident entry points beyond 128 states
        es.close()


@pytest.mark.parametrize("n", [129, 150, 192, 193, 200, 208, 209, 224, 225, 256, 300, 513, 1024])   # register kernels: C = 96 (.. 192), 104 (.. 208), 112 (.. 224); beyond: the matrix from the L2
def test_wide_vs_oracle_random(hip, oracle, n):
    """Random dense HMMs, a bootstrap multiset with repeats, edge lengths; decoding (-d, -D, -c branches) on the resident tables."""
    rng = np.random.default_rng(500 + n)
    a, e, a0 = random_hmm(rng, n)
    lens = (1, 2, 63, 64, 65, 129, 400) if n <= 513 else (1, 2, 65, 130)
    segs = [rng.choice(3, size=L, p=[0.86, 0.1, 0.04]).astype(np.uint8) for L in lens]
    sel = [len(segs) - 1, 0, 3, len(segs) - 1, 2, 1]
    es = hip.HipEStep(n, mode=hip.MODE_EXACT)
    es.load_segments(segs)
    es.select(sel)
    r = es.estep(a, e, a0)
    o = oracle.estep(a, e, a0, [segs[i] for i in sel], per_seg=True)
    assert bits_equal(r["A"], o["A"]) and bits_equal(r["E"], o["E"]) and bits_equal(r["A0"], o["A0"])
    assert r["LL"] == o["LL"]
    assert bits_equal(r["chk"], o["seg_chk"])
    k = len(segs) - 1
    f, b, s, lk, chk = oracle.fwd_bwd(a, e, a0, segs[k])
    path, mp = oracle.post_decode(f, b, s)
    gp, gm = es.decode(k)
    assert np.array_equal(gp, path[1:]) and bits_equal(gm, mp[1:])
    post, rec = oracle.post_full(a, e, segs[k], f, b, s)
    pp, rr = es.posterior(k)
    assert bits_equal(pp, post[1:]) and bits_equal(rr, rec[1:])
    c1 = rng.integers(0, 50, size=(len(segs[k]) - 3, 4), dtype=np.int32)
    cd = np.zeros((n, 4)); co = np.zeros((n, 4))
    oracle.post_counts(f, b, s, c1, co); es.post_counts(k, c1, cd)
    oracle.post_counts(f, b, s, c1, co); es.post_counts(k, c1, cd)   # running totals carried over
    assert bits_equal(cd, co)
    es.close()


@pytest.mark.parametrize("sort", [1, 0])
def test_wide_batch_equals_separate_calls(hip, oracle, sort):
    """psmc_hip_estep_batch at 150 states: the general entry scheduler with one entry per work-group."""
    rng = np.random.default_rng(77)
    n = 150
    segs = [rng.choice(3, size=L, p=[0.86, 0.1, 0.04]).astype(np.uint8) for L in (1, 5, 64, 65, 200, 333, 90)]
    pars = [random_hmm(rng, n) for _ in range(3)]
    sels = [[6, 0, 3, 6, 5], [4, 4, 2], [1, 2, 3, 4, 5, 6, 0]]
    es = hip.HipEStep(n, mode=hip.MODE_EXACT, batch_sort=sort, batch_bins=600)
    es.load_segments(segs)
    got = es.estep_batch(pars, sels)
    assert es.batch_info()["groups"] >= 2
    for r, sel in enumerate(sels):
        o = oracle.estep(pars[r][0], pars[r][1], pars[r][2], [segs[i] for i in sel])
        assert bits_equal(got["A"][r], o["A"]) and bits_equal(got["E"][r], o["E"]) and got["LL"][r] == o["LL"], r
    es.close()


def test_wide_general_kernels_at_register_kernel_sizes():
    """129 .. 224 states run the register-resident kernels (k_fwd_wide2 / k_bwd_wide2) by default; PSMC_HIP_WIDE_L2=1 sends the same sizes
    through the general kernels (matrix from the L2, what 225 .. 1024 states use): both give the reference's bits on the 200- and
    149-state goldens."""
    import subprocess, sys
    code = """
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from psmc_amd import hip
from conftest import bits_equal, Golden
g = dict(np.load(os.path.join(%r, "tests", "golden", "estep_wide.npz")))
segs = Golden().segs_small[:8]
for key in ("n200", "n149"):
    a, e, a0 = g[key + ".a"], g[key + ".e"], g[key + ".a0"]
    es = hip.HipEStep(a.shape[0], mode=hip.MODE_EXACT); es.load_segments(segs)
    r = es.estep(a, e, a0)
    assert bits_equal(r["A"], g[key + ".A"]) and bits_equal(r["E"], g[key + ".E"]) and r["LL"] == float(g[key + ".LL"]) and bits_equal(r["chk"], g[key + ".seg_chk"]), key
    f, b, sc = es.tables(5)
    assert bits_equal(f[::7], g[key + ".f65"]) and bits_equal(b[::7], g[key + ".b65"]) and bits_equal(sc, g[key + ".s65"]), key
    es.close()
print("wide l2 ok")
""" % (ROOT, ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PSMC_HIP_WIDE_L2="1"), timeout=600)
    assert r.returncode == 0 and "wide l2 ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
