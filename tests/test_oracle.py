"""The CPU oracle (oracle/psmc_oracle.c) against the golden vectors dumped from
the real reference, bit for bit, and against the reference itself when its
checkout is present.  Pins the oracle."""
import numpy as np
import pytest
from conftest import bits_equal


@pytest.mark.parametrize("key", ["n64_flat", "n64_curve", "n23_flat", "n23_curve"])
def test_estep_small_bitexact(golden, oracle, key):
    p = golden.params(key)
    r = oracle.estep(p["a"], p["e"], p["a0"], golden.segs_small, per_seg=True)
    g = golden.small
    assert bits_equal(r["A"], g[key + ".A"])
    assert bits_equal(r["E"], g[key + ".E"])
    assert bits_equal(r["A0"], g[key + ".A0"])
    assert r["LL"] == float(g[key + ".LL"])
    assert bits_equal(r["seg_E"], g[key + ".seg_E"])
    assert bits_equal(r["seg_LL"], g[key + ".seg_LL"])
    assert bits_equal(r["seg_chk"], g[key + ".seg_chk"])
    assert bits_equal(r["seg_A"][[0, 5, 9]], g[key + ".seg_A_pick"])
    assert bits_equal(r["seg_A"].sum(2), g[key + ".seg_A_rowsum"])


@pytest.mark.parametrize("key", ["n64_curve", "n23_flat"])
def test_tables_bitexact(golden, oracle, key):
    p = golden.params(key)
    f, b, s, lk, chk = oracle.fwd_bwd(p["a"], p["e"], p["a0"], golden.segs_small[5])
    g = golden.small
    assert bits_equal(f, g[key + ".f65"]) and bits_equal(b, g[key + ".b65"]) and bits_equal(s, g[key + ".s65"])
    assert lk == float(g[key + ".lk65"])


def test_estep_n128_bitexact(golden, oracle):
    """-p "64*2" (128 states): E-step statistics and tables of the reference."""
    g, k = golden.n128, "n128_curve"
    a, e, a0 = g[k + ".a"], g[k + ".e"], g[k + ".a0"]
    assert a.shape == (128, 128)
    r = oracle.estep(a, e, a0, golden.segs_small, per_seg=True)
    assert bits_equal(r["A"], g[k + ".A"]) and bits_equal(r["E"], g[k + ".E"]) and bits_equal(r["A0"], g[k + ".A0"])
    assert r["LL"] == float(g[k + ".LL"])
    assert bits_equal(r["seg_E"], g[k + ".seg_E"]) and bits_equal(r["seg_LL"], g[k + ".seg_LL"])
    assert bits_equal(r["seg_chk"], g[k + ".seg_chk"])
    assert bits_equal(r["seg_A"].sum(2), g[k + ".seg_A_rowsum"]) and bits_equal(r["seg_A"].sum(1), g[k + ".seg_A_colsum"])
    f, b, s, lk, chk = oracle.fwd_bwd(a, e, a0, golden.segs_small[5])
    assert bits_equal(f, g[k + ".f65"]) and bits_equal(b, g[k + ".b65"]) and bits_equal(s, g[k + ".s65"])
    assert lk == float(g[k + ".lk65"])


@pytest.mark.parametrize("key", ["n64_flat", "n64_curve"])
def test_estep_mid_bitexact(golden, oracle, key):
    p = golden.params(key)
    r = oracle.estep(p["a"], p["e"], p["a0"], golden.segs_mid, per_seg=True)
    g = golden.mid
    assert bits_equal(r["A"], g[key + ".A"]) and bits_equal(r["E"], g[key + ".E"])
    assert r["LL"] == float(g[key + ".LL"])
    assert bits_equal(r["seg_LL"], g[key + ".seg_LL"]) and bits_equal(r["seg_chk"], g[key + ".seg_chk"])


@pytest.mark.parametrize("key", ["n200", "n149"])
def test_estep_wide_bitexact(golden, oracle, key):
    """Beyond 128 states (`-p "100*2"`: 200; `-p "4+47*3+4"`: 149): the reference's statistics and thinned tables
    (tests/golden/make_golden_wide.py) -- the pin for the wide exact kernels of round 5."""
    import os
    from conftest import GOLD
    g = dict(np.load(os.path.join(GOLD, "estep_wide.npz")))
    a, e, a0 = g[key + ".a"], g[key + ".e"], g[key + ".a0"]
    r = oracle.estep(a, e, a0, golden.segs_small[:8], per_seg=True)
    assert bits_equal(r["A"], g[key + ".A"]) and bits_equal(r["E"], g[key + ".E"]) and bits_equal(r["A0"], g[key + ".A0"])
    assert r["LL"] == float(g[key + ".LL"]) and bits_equal(r["seg_chk"], g[key + ".seg_chk"]) and bits_equal(r["seg_E"], g[key + ".seg_E"])
    f, b, s, lk, chk = oracle.fwd_bwd(a, e, a0, golden.segs_small[5])
    assert bits_equal(f[1::7], g[key + ".f65"]) and bits_equal(b[1::7], g[key + ".b65"]) and bits_equal(s[1:], g[key + ".s65"])


def test_estep_stress_bitexact(oracle):
    """The real-data-shaped fixture (N runs of 2e4 / 2e5 bins, a run of homozygosity, a het-dense stretch, a segment that is
    one long N run; tests/golden/make_golden_stress.py): the reference's statistics at the parameters of EM round 1."""
    import gzip, os
    from conftest import GOLD
    lut = np.full(256, 2, np.uint8); lut[ord("T")] = 0; lut[ord("K")] = 1
    segs, cur = [], []
    for line in gzip.open(os.path.join(GOLD, "stress", "stress.psmcfa.gz"), "rb"):
        if line.startswith(b">"):
            if cur: segs.append(np.concatenate(cur))
            cur = []
        else:
            cur.append(lut[np.frombuffer(line.rstrip(b"\n"), dtype=np.uint8)])
    segs.append(np.concatenate(cur))
    g = dict(np.load(os.path.join(GOLD, "stress", "stress_estep.npz")))
    r = oracle.estep(g["rd1.a"], g["rd1.e"], g["rd1.a0"], segs, per_seg=True)
    assert bits_equal(r["A"], g["rd1.A"]) and bits_equal(r["E"], g["rd1.E"]) and r["LL"] == float(g["rd1.LL"])
    assert bits_equal(r["seg_LL"], g["rd1.seg_LL"]) and bits_equal(r["seg_chk"], g["rd1.seg_chk"])


def test_invariants(golden, oracle):
    """sum A = sum E = L-1 per segment (+ the HMM_TINY seeds); posterior sums to 1."""
    p = golden.params("n64_curve")
    r = oracle.estep(p["a"], p["e"], p["a0"], golden.segs_small, per_seg=True)
    for i, seg in enumerate(golden.segs_small):
        L = len(seg)
        assert abs(r["seg_A"][i].sum() - (L - 1)) < 1e-8 * max(L, 1) + 1e-20
        assert abs(r["seg_E"][i].sum() - (L - 1)) < 1e-8 * max(L, 1) + 1e-20
        assert abs(r["seg_chk"][i] - 1.0) < 1e-9
    f, b, s, lk, chk = oracle.fwd_bwd(p["a"], p["e"], p["a0"], golden.segs_small[8])
    post = f[1:] * b[1:] * s[1:, None]
    assert np.allclose(post.sum(1), 1.0, atol=1e-10)


def test_Q_functions(golden, oracle):
    """hmm_Q0/hmm_Q (khmm.c:326-382) against the reference's EM round: Q0 printed by psmc_em is
    hmm_Q at the current parameters."""
    p = golden.params("n64_flat")
    g = golden.mid
    A, E = g["n64_flat.A"], g["n64_flat.E"]
    q0 = oracle.Q0(A, E)
    q = oracle.Q(p["a"], p["e"], A, E, q0)
    assert q == float(g["em_n64_flat.Q0"])


def test_oracle_vs_reference_random(reference, oracle):
    rng = np.random.default_rng(5)
    for n in (3, 23, 64):
        a = rng.random((n, n)) ** 3 + np.eye(n) * 5
        a /= a.sum(1, keepdims=True)
        e = np.ones((3, n)); e[1] = rng.random(n) * 0.2; e[0] = 1 - e[1]
        a0 = rng.random(n); a0 /= a0.sum()
        segs = [rng.choice(3, size=L, p=[0.85, 0.1, 0.05]).astype(np.uint8) for L in (1, 2, 17, 500, 3000)]
        ro = oracle.estep(a, e, a0, segs, per_seg=True)
        rr = reference.estep(a, e, a0, segs, per_seg=True)
        for k in ro:
            assert bits_equal(np.asarray(ro[k]), np.asarray(rr[k])), (n, k)


def test_oracle_decode_branches_match_plain_restatement(oracle, golden):
    """orc_post_full / orc_post_counts (aux.c:183-231) against a loop-for-loop Python restatement (IEEE doubles, no
    FMA, same order) on a 65-bin segment; the CLI goldens small_decode_D / small_decode_c pin the same code against the
    reference binary's printed output (tests/test_host_cli.py)."""
    p = golden.params("n23_flat")
    a, e, a0 = p["a"], p["e"], p["a0"]
    seg = golden.segs_small[5]
    f, b, s, lk, chk = oracle.fwd_bwd(a, e, a0, seg)
    post, rec = oracle.post_full(a, e, seg, f, b, s)
    L, n = len(seg), a.shape[0]
    rng = np.random.default_rng(3)
    c1 = rng.integers(0, 30, size=(L - 2, 3), dtype=np.int32)
    cnt = np.zeros((n, 3)); want = [[0.0] * 3 for _ in range(n)]
    oracle.post_counts(f, b, s, c1, cnt)
    for k in range(1, L + 1):
        if k < L:
            pr = 0.0
            for l in range(n):
                pr += float(f[k][l]) * float(a[l][l]) * float(b[k + 1][l]) * float(e[seg[k]][l])
            pr = 1.0 - pr
        else:
            pr = 0.0
        assert pr == rec[k]
        for l in range(n):
            q = float(f[k][l]) * float(b[k][l]) * float(s[k])
            assert q == post[k][l]
            if k <= L - 2:
                for j in range(3):
                    want[l][j] += q * int(c1[k - 1][j])
    assert np.array_equal(cnt, np.array(want))
