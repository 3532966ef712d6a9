"""ctypes access to the CPU oracle (oracle/libpsmc_oracle.so) and, when it has
been built in this container, to the real reference (oracle/_ref/libpsmc_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package psmc_amd.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_SO = os.path.join(ORC_DIR, "libpsmc_oracle.so")
REF_SO = os.path.join(ORC_DIR, "_ref", "libpsmc_ref.so")
REF_BIN = os.path.join(ORC_DIR, "_ref", "psmc_ref")

c_dp = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)


def build_oracle(with_ref=None):
    """Compile the oracle (and oracle/_ref when the reference checkout exists)."""
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference")
    targets = ["all"] + (["ref"] if with_ref else [])
    subprocess.run(["make", "-s", "-C", ORC_DIR] + targets, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _dp(x):
    return x.ctypes.data_as(c_dp) if x is not None else None


def _seg_args(segs):
    segs = [np.ascontiguousarray(s, dtype=np.uint8) for s in segs]
    n = len(segs)
    ptrs = (c_u8p * n)(*[s.ctypes.data_as(c_u8p) for s in segs])
    lens = np.array([len(s) for s in segs], dtype=np.int32)
    return segs, ptrs, lens


class _Lib:
    prefix = ""

    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def estep(self, a, e, a0, segs, per_seg=False):
        """Returns dict(A, E (2,n), A0, LL[, seg_A, seg_E (n_seg,3,n), seg_LL, seg_chk])."""
        n = a.shape[0]
        a = np.ascontiguousarray(a, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        assert e.shape == (3, n)
        segs, ptrs, lens = _seg_args(segs)
        ns = len(segs)
        A = np.zeros((n, n)); E = np.zeros((2, n)); A0 = np.zeros(n); LL = C.c_double(0)
        sA = np.zeros((ns, n, n)) if per_seg else None
        sE = np.zeros((ns, 3, n)) if per_seg else None
        sL = np.zeros(ns) if per_seg else None
        sC = np.zeros(ns) if per_seg else None
        f = self._fn("estep")
        f.restype = None if self.prefix == "orc_" else C.c_int
        f(C.c_int(n), _dp(a), _dp(e), _dp(a0), C.c_int(ns), ptrs,
          lens.ctypes.data_as(c_i32p), _dp(A), _dp(E), _dp(A0), C.byref(LL),
          _dp(sA), _dp(sE), _dp(sL), _dp(sC))
        out = dict(A=A, E=E, A0=A0, LL=LL.value)
        if per_seg:
            out.update(seg_A=sA, seg_E=sE, seg_LL=sL, seg_chk=sC)
        return out


class Oracle(_Lib):
    prefix = "orc_"

    def __init__(self):
        if not os.path.exists(ORC_SO):
            build_oracle(with_ref=False)
        super().__init__(ORC_SO)

    def fwd_bwd(self, a, e, a0, seg):
        """f, b ((L+1, n), row 0 unused), s (L+1), lk, chk -- khmm.c:145-260."""
        n = a.shape[0]
        a = np.ascontiguousarray(a, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        seg = np.ascontiguousarray(seg, dtype=np.uint8)
        L = len(seg)
        f = np.zeros((L + 1, n)); b = np.zeros((L + 1, n)); s = np.zeros(L + 1)
        ae = np.zeros((3, n, n))
        self.lib.orc_pre_backward(C.c_int(n), _dp(a), _dp(e), _dp(ae))
        self.lib.orc_forward(C.c_int(n), _dp(a), _dp(e), _dp(a0), C.c_int(L),
                             seg.ctypes.data_as(c_u8p), _dp(f), _dp(s))
        self.lib.orc_backward.restype = C.c_double
        chk = self.lib.orc_backward(C.c_int(n), _dp(ae), _dp(e), _dp(a0), C.c_int(L),
                                    seg.ctypes.data_as(c_u8p), _dp(s), _dp(b))
        self.lib.orc_lk.restype = C.c_double
        lk = self.lib.orc_lk(C.c_int(L), _dp(s))
        return f, b, s, lk, chk

    def post_decode(self, f, b, s):
        L, n = f.shape[0] - 1, f.shape[1]
        path = np.zeros(L + 1, dtype=np.int32); mp = np.zeros(L + 1)
        self.lib.orc_post_decode(C.c_int(n), C.c_int(L), _dp(f), _dp(b), _dp(s),
                                 path.ctypes.data_as(c_i32p), _dp(mp))
        return path, mp

    def post_full(self, a, e, seg, f, b, s):
        """post ((L+1, n)), recomb (L+1,) of aux.c:183-200; row / entry 0 unused."""
        L, n = f.shape[0] - 1, f.shape[1]
        seg = np.ascontiguousarray(seg, dtype=np.uint8)
        post = np.zeros((L + 1, n)); rec = np.zeros(L + 1)
        self.lib.orc_post_full(C.c_int(n), _dp(np.ascontiguousarray(a)), _dp(np.ascontiguousarray(e)), C.c_int(L),
                               seg.ctypes.data_as(c_u8p), _dp(f), _dp(b), _dp(s), _dp(post), _dp(rec))
        return post, rec

    def post_counts(self, f, b, s, cnt1, cnt):
        """cnt (n, n_cnt) += posterior-weighted counts (aux.c:202-219); in place."""
        L, n = f.shape[0] - 1, f.shape[1]
        cnt1 = np.ascontiguousarray(cnt1, dtype=np.int32)
        self.lib.orc_post_counts(C.c_int(n), C.c_int(L), _dp(f), _dp(b), _dp(s), cnt1.ctypes.data_as(c_i32p),
                                 C.c_int32(cnt1.shape[0]), C.c_int32(cnt1.shape[1]), _dp(cnt))
        return cnt

    def Q0(self, A, E):
        self.lib.orc_Q0.restype = C.c_double
        return self.lib.orc_Q0(C.c_int(A.shape[0]), _dp(np.ascontiguousarray(A)),
                               _dp(np.ascontiguousarray(E)))

    def Q(self, a, e, A, E, Q0):
        self.lib.orc_Q.restype = C.c_double
        return self.lib.orc_Q(C.c_int(A.shape[0]), _dp(np.ascontiguousarray(a)),
                              _dp(np.ascontiguousarray(e)), _dp(np.ascontiguousarray(A)),
                              _dp(np.ascontiguousarray(E)), C.c_double(Q0))


class Reference(_Lib):
    """The real lh3/psmc objects behind oracle/ref_shim.c (container-only)."""
    prefix = "ref_"

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        super().__init__(REF_SO)

    def fwd_bwd(self, a, e, a0, seg):
        n = a.shape[0]
        a = np.ascontiguousarray(a, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        seg = np.ascontiguousarray(seg, dtype=np.uint8)
        L = len(seg)
        f = np.zeros((L + 1, n)); b = np.zeros((L + 1, n)); s = np.zeros(L + 1)
        lk = C.c_double(0)
        # hmm_new_data copies seq into a 1-indexed buffer itself (khmm.c:42-43)
        self.lib.ref_fwd_bwd(C.c_int(n), _dp(a), _dp(e), _dp(a0), C.c_int(L),
                             seg.ctypes.data_as(c_u8p), _dp(f), _dp(b), _dp(s), C.byref(lk))
        return f, b, s, lk.value

    def parse_pattern(self, pattern):
        pm = np.zeros(512, dtype=np.int32); nf = C.c_int(0)
        n = self.lib.ref_parse_pattern(pattern.encode(), C.byref(nf), pm.ctypes.data_as(c_i32p))
        return n, nf.value, pm[: n + 1].copy()

    def hmm_params(self, pattern, params, alpha=0.1, dt0=-1.0):
        n, nf, _ = self.parse_pattern(pattern)
        N = n + 1
        params = np.ascontiguousarray(params, dtype=np.float64)
        t = np.zeros(n + 2); a = np.zeros((N, N)); e = np.zeros((3, N)); a0 = np.zeros(N)
        sg = np.zeros(N); cpi = C.c_double(0); csg = C.c_double(0)
        self.lib.ref_hmm_params(pattern.encode(), _dp(params), C.c_double(alpha), C.c_double(dt0),
                                _dp(t), _dp(a), _dp(e), _dp(a0), _dp(sg), C.byref(cpi), C.byref(csg))
        return dict(t=t, a=a, e=e, a0=a0, sigma=sg, C_pi=cpi.value, C_sigma=csg.value)

    def em_round(self, pattern, params, segs, alpha=0.1, dt0=-1.0):
        n, nf, _ = self.parse_pattern(pattern)
        p = np.array(params, dtype=np.float64).copy()
        segs, ptrs, lens = _seg_args(segs)
        lk = C.c_double(0); q0 = C.c_double(0); q1 = C.c_double(0)
        ps = np.zeros(n + 1)
        it = self.lib.ref_em_round(pattern.encode(), _dp(p), C.c_double(alpha), C.c_double(dt0),
                                   C.c_int(len(segs)), ptrs, lens.ctypes.data_as(c_i32p),
                                   C.byref(lk), C.byref(q0), C.byref(q1), _dp(ps))
        return dict(params=p, lk=lk.value, Q0=q0.value, Q1=q1.value, post_sigma=ps, IT=it)

    def resample(self, seed, lens):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        out = np.zeros(4 * len(lens) + 16, dtype=np.int32)
        m = self.lib.ref_resample(C.c_long(seed), C.c_int(len(lens)), lens.ctypes.data_as(c_i32p),
                                  out.ctypes.data_as(c_i32p), C.c_int(len(out)))
        return out[:m].copy()

    def read_psmcfa(self, fn, max_seg=4096):
        L = np.zeros(max_seg, dtype=np.int32); Le = L.copy(); ne = L.copy()
        sL = C.c_int64(0); sn = C.c_int(0)
        m = self.lib.ref_read_psmcfa(fn.encode(), C.c_int(max_seg), L.ctypes.data_as(c_i32p),
                                     Le.ctypes.data_as(c_i32p), ne.ctypes.data_as(c_i32p),
                                     None, C.byref(sL), C.byref(sn))
        buf = np.zeros(int(L[:m].sum()), dtype=np.uint8)
        self.lib.ref_read_psmcfa(fn.encode(), C.c_int(max_seg), L.ctypes.data_as(c_i32p),
                                 Le.ctypes.data_as(c_i32p), ne.ctypes.data_as(c_i32p),
                                 buf.ctypes.data_as(c_u8p), C.byref(sL), C.byref(sn))
        off = np.concatenate([[0], np.cumsum(L[:m])])
        segs = [buf[off[i]:off[i + 1]].copy() for i in range(m)]
        return dict(segs=segs, L=L[:m].copy(), L_e=Le[:m].copy(), n_e=ne[:m].copy(),
                    sum_L=sL.value, sum_n=sn.value)

    def kmin_quad(self, x, centre):
        x = np.array(x, dtype=np.float64).copy(); c = np.ascontiguousarray(centre, dtype=np.float64)
        self.lib.ref_kmin_quad.restype = C.c_double
        fx = self.lib.ref_kmin_quad(C.c_int(len(x)), _dp(x), _dp(c))
        return x, fx


def have_reference():
    return os.path.exists(REF_SO)
