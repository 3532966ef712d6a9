"""The numpy specification of the fast-mode algorithm (tests/fastmodel.py)
against the oracle: checks the algebra (lagged normalisation, shared divisors,
per-tile posterior normalisation, a .* C factorisation) and documents how the
error depends on the warm-up length W."""
import numpy as np
import pytest
import fastmodel


def relmax(x, y):
    return float(np.abs(x - y).max() / np.abs(y).max())


def test_untiled_matches_oracle(golden, oracle):
    p = golden.params("n64_curve")
    segs = golden.segs_small
    ref = oracle.estep(p["a"], p["e"], p["a0"], segs)
    m = fastmodel.estep_fast_model(p["a"], p["e"], p["a0"], segs, T=1 << 30, W=0)
    assert relmax(m["A"], ref["A"]) < 1e-12 and relmax(m["E"], ref["E"]) < 1e-12
    assert abs(m["LL"] - ref["LL"]) < 1e-12 * abs(ref["LL"])


@pytest.mark.parametrize("T,W,tol", [(1024, 4096, 1e-11), (4096, 4096, 1e-11), (512, 8192, 1e-12)])
def test_tiled_matches_oracle(golden, oracle, T, W, tol):
    p = golden.params("n64_curve")
    segs = golden.segs_mid[2:]   # 20000, 12000, 5000, 800 bins
    ref = oracle.estep(p["a"], p["e"], p["a0"], segs)
    m = fastmodel.estep_fast_model(p["a"], p["e"], p["a0"], segs, T=T, W=W)
    assert relmax(m["A"], ref["A"]) < tol and relmax(m["E"], ref["E"]) < tol
    assert abs(m["LL"] - ref["LL"]) < tol * abs(ref["LL"])


def test_short_overlap_alone_is_wrong_and_repair_fixes_it(golden, oracle):
    """An overlap far below the forgetting length must NOT pass by itself (this is what the
    verify kernel of the HIP path detects); with verify + repair it is right again, at a
    fraction of the work a long fixed overlap would cost."""
    p = golden.params("n64_curve")
    segs = golden.segs_mid[2:4]
    ref = oracle.estep(p["a"], p["e"], p["a0"], segs)
    m = fastmodel.estep_fast_model(p["a"], p["e"], p["a0"], segs, T=1024, W=128)
    assert relmax(m["A"], ref["A"]) > 1e-6
    st = {}
    m = fastmodel.estep_fast_model(p["a"], p["e"], p["a0"], segs, T=1024, W=128, tol=1e-12, stats=st)
    assert relmax(m["A"], ref["A"]) < 1e-11 and relmax(m["E"], ref["E"]) < 1e-11
    assert abs(m["LL"] - ref["LL"]) < 1e-12 * abs(ref["LL"])
    assert st["fwd_rounds"] >= 1 and st["fwd_steps"] < 4 * st["bins"]


def test_no_overlap_with_repair(golden, oracle):
    p = golden.params("n64_flat")
    segs = golden.segs_small
    ref = oracle.estep(p["a"], p["e"], p["a0"], segs)
    m = fastmodel.estep_fast_model(p["a"], p["e"], p["a0"], segs, T=500, W=0, tol=1e-12)
    assert relmax(m["A"], ref["A"]) < 1e-11 and relmax(m["E"], ref["E"]) < 1e-11


def test_structured_factorisation(golden):
    """The O(N) sweeps rest on a[k][l] = P_k qa_l below and R_k c_l above the diagonal (core.c:112-122):
    every golden transition matrix factors to 64 ulp with dd >= 0, the two-scan step equals the dense
    product, and a capped matrix (aux.c:115-127) or a generic one is rejected (dense fallback)."""
    rng = np.random.default_rng(3)
    for key in golden.param_keys() + ["n128"]:
        a = golden.n128["n128_curve.a"] if key == "n128" else golden.params(key)["a"]
        f = fastmodel.factor_structure(a)
        assert f is not None, key
        x = rng.random(a.shape[0]) ** 3
        assert relmax(fastmodel.struct_step_forward(f, x), a.T @ x) < 1e-13
        assert relmax(fastmodel.struct_step_backward(f, x), a @ x) < 1e-13
    a = golden.params("n64_curve")["a"].copy()
    a[:, 40] = a[:, 40:].sum(1); a[:, 41:] = 0.0
    assert fastmodel.factor_structure(a) is None
    b = rng.random((64, 64)); b /= b.sum(1, keepdims=True)
    assert fastmodel.factor_structure(b) is None
