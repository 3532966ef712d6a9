"""ctypes mirror of the host driver library (psmc_amd/host/libpsmc_host.so): the PSMC model -> HMM parameter map
(psmc_update_hmm, lh3/psmc core.c:61-133) for tools that need (a, e, a0) for a `PA` parameter vector -- bench.py's
parameter trajectory, scripts.  Plain host C, no GPU code."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Pattern(C.Structure):
    _fields_ = [("n_states", C.c_int), ("n_free", C.c_int), ("group", C.POINTER(C.c_int))]


class _Model(C.Structure):  # psmc_model of psmc_amd/host/psmc_host.h, field for field
    _fields_ = [("pat", _Pattern), ("pattern_text", C.c_char_p), ("alpha", C.c_double), ("has_dt", C.c_int),
                ("fixed_t", C.POINTER(C.c_double)), ("n_params", C.c_int), ("params", C.POINTER(C.c_double)),
                ("t", C.POINTER(C.c_double)), ("sigma", C.POINTER(C.c_double)), ("post_sigma", C.POINTER(C.c_double)),
                ("C_pi", C.c_double), ("C_sigma", C.c_double),
                ("a", C.POINTER(C.c_double)), ("e", C.POINTER(C.c_double)), ("a0", C.POINTER(C.c_double)),
                ("lk", C.c_double), ("Q0", C.c_double), ("Q1", C.c_double), ("fast_mstep", C.c_int)]


def load():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "host", "libpsmc_host.so")
        if not os.path.exists(p):
            raise RuntimeError("%s not built: run `make -C psmc_amd/host`" % p)
        lib = C.CDLL(p)
        lib.psmc_pattern_parse.argtypes = [C.c_char_p, C.POINTER(_Pattern)]
        lib.psmc_pattern_free.argtypes = [C.POINTER(_Pattern)]
        lib.psmc_model_new.restype = C.POINTER(_Model)
        lib.psmc_model_new.argtypes = [C.POINTER(_Pattern), C.c_char_p, C.c_double, C.c_int]
        lib.psmc_model_update.argtypes = [C.POINTER(_Model)]
        lib.psmc_model_free.argtypes = [C.POINTER(_Model)]
        _LIB = lib
    return _LIB


def hmm_params(pattern, params, alpha=0.1):
    """(a (N,N), e (3,N), a0 (N,)) of the model with PA-line parameters [theta0, rho0, max_t, lambda_0 ..]."""
    lib = load()
    pat = _Pattern()
    if lib.psmc_pattern_parse(pattern.encode(), C.byref(pat)) != 0:
        raise ValueError("malformed pattern %r" % pattern)
    m = lib.psmc_model_new(C.byref(pat), pattern.encode(), alpha, 0)
    try:
        N, npar = pat.n_states, m.contents.n_params
        params = np.asarray(params, dtype=np.float64)
        if len(params) != npar:
            raise ValueError("pattern %s takes %d parameters, got %d" % (pattern, npar, len(params)))
        for i in range(npar):
            m.contents.params[i] = float(params[i])
        lib.psmc_model_update(m)
        a = np.ctypeslib.as_array(m.contents.a, shape=(N, N)).copy()
        e = np.ctypeslib.as_array(m.contents.e, shape=(3, N)).copy()
        a0 = np.ctypeslib.as_array(m.contents.a0, shape=(N,)).copy()
    finally:
        lib.psmc_model_free(m)
        lib.psmc_pattern_free(C.byref(pat))
    return a, e, a0
