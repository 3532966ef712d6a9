"""ctypes mirror of include/psmc_hip.h.

`HipEStep` plays the role the khmm.h call sequence plays inside psmc_em()
(lh3/psmc em.c:33-55): give it the segments once, then per EM iteration hand it
the HMM parameters (a, e, a0) and get back the summed expected counts
he_sum->A, he_sum->E[0..1] and the log-likelihood.  There is NO CPU fallback:
if libpsmc_hip.so is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os
import numpy as np

MODE_EXACT = 0
MODE_FAST = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_DIAG = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

# every symbol include/psmc_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "psmc_hip_device_count", "psmc_hip_device_cus", "psmc_hip_set_cu_range", "psmc_hip_reserve_tables", "psmc_hip_create", "psmc_hip_destroy", "psmc_hip_strerror",
    "psmc_hip_last_error", "psmc_hip_set_option", "psmc_hip_load_segments",
    "psmc_hip_load_segments_device", "psmc_hip_select", "psmc_hip_estep",
    "psmc_hip_estep_segments", "psmc_hip_estep_batch", "psmc_hip_estep_batch_cb", "psmc_hip_reserve_batch_tables", "psmc_hip_batch_info", "psmc_hip_estep_device", "psmc_hip_fast_diag", "psmc_hip_fast_repairs", "psmc_hip_fast_info", "psmc_hip_estep_factored",
    "psmc_hip_get_tables", "psmc_hip_decode", "psmc_hip_posterior", "psmc_hip_post_counts",
    "psmc_hip_group_selfcheck", "psmc_hip_fast_plan",
    "psmc_hip_group_create", "psmc_hip_group_destroy", "psmc_hip_group_last_error", "psmc_hip_group_set_option",
    "psmc_hip_group_load_segments", "psmc_hip_group_estep", "psmc_hip_group_estep_factored", "psmc_hip_group_info",
    "psmc_hip_group_route", "psmc_hip_estep_factored_device",
]

# every symbol include/psmc_hip_diag.h declares: libpsmc_hip_diag.so, the lab bench -- not part of the drop-in library
DIAG_EXPORTS = [
    "psmc_hip_selftest", "psmc_hip_last_timing", "psmc_hip_microbench", "psmc_hip_stream_probe", "psmc_hip_hbm_probe",
    "psmc_hip_load_probe", "psmc_hip_load_probe_st", "psmc_hip_pipe_probe", "psmc_hip_pipe_probe2", "psmc_hip_place_probe",
    "psmc_hip_cumask_probe",
]


class HipError(RuntimeError):
    pass


def lib_path():
    """psmc_amd/libpsmc_hip.so; PSMC_HIP_LIB names another build of the same library (A/B timing of a kernel variant)."""
    return os.environ.get("PSMC_HIP_LIB") or os.path.join(_HERE, "libpsmc_hip.so")


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, like /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed make sure ITS runtime is
    the one mapped before libpsmc_hip.so resolves libamdhip64.so.7 -- then torch tensors, torch streams and
    our kernels live in one runtime whichever is imported first.  Without torch (e.g. the C driver) the
    system runtime under /opt/rocm is used."""
    import sys
    if "torch" in sys.modules or os.environ.get("PSMC_HIP_SYSTEM_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see api.hip: streams of one E-step must not share a hardware queue


def load_library():
    """dlopen psmc_amd/libpsmc_hip.so (built by psmc_amd/csrc/Makefile).  Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise HipError("%s not built: run `make -C psmc_amd/csrc` (or __graft_entry__.build())" % p)
    _share_torch_hip_runtime()
    lib = C.CDLL(p)
    lib.psmc_hip_strerror.restype = C.c_char_p
    lib.psmc_hip_last_error.restype = C.c_char_p
    lib.psmc_hip_last_error.argtypes = [C.c_void_p]
    lib.psmc_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    lib.psmc_hip_destroy.argtypes = [C.c_void_p]
    lib.psmc_hip_destroy.restype = None
    lib.psmc_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.psmc_hip_load_segments.argtypes = [C.c_void_p, C.c_int, C.POINTER(_u8p), _i32p]
    lib.psmc_hip_load_segments_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _i64p, _i32p]
    lib.psmc_hip_select.argtypes = [C.c_void_p, C.c_int, _i32p]
    lib.psmc_hip_set_cu_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.psmc_hip_reserve_tables.argtypes = [C.c_void_p]
    lib.psmc_hip_device_cus.argtypes = [C.c_int]
    lib.psmc_hip_estep.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.psmc_hip_estep_segments.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.psmc_hip_estep_device.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_void_p, C.c_void_p]
    lib.psmc_hip_fast_diag.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.psmc_hip_fast_repairs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.psmc_hip_fast_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.psmc_hip_get_tables.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    lib.psmc_hip_decode.argtypes = [C.c_void_p, C.c_int, _i32p, _dp]
    _LIB = lib
    return lib


def load_diag():
    """dlopen psmc_amd/libpsmc_hip_diag.so (self-test, probes, event timing; include/psmc_hip_diag.h).  It links against
    libpsmc_hip.so, which is loaded first so that both resolve the same copy."""
    global _DIAG
    if _DIAG is not None:
        return _DIAG
    load_library()
    p = os.path.join(os.path.dirname(lib_path()), "libpsmc_hip_diag.so")
    if not os.path.exists(p):
        raise HipError("%s not built: run `make -C psmc_amd/csrc`" % p)
    d = C.CDLL(p)
    d.psmc_hip_selftest.argtypes = [C.c_int]
    d.psmc_hip_last_timing.argtypes = [C.c_void_p, _dp]
    _DIAG = d
    return d


def _p(x):
    return x.ctypes.data_as(_dp) if x is not None else None


class HipEStep:
    """One E-step engine bound to one GPU.

    a: (n, n) transition matrix, a[k, l] = P(k -> l)       (khmm.h:34)
    e: (2, n) or (3, n) emission rows hom / het [/ missing] (khmm.h:34, khmm.c:21)
    a0: (n,) initial distribution                           (khmm.h:36)
    """

    def __init__(self, n_states, device=0, mode=MODE_FAST, **options):
        self.lib = load_library()
        if self.lib.psmc_hip_device_count() <= 0:
            raise HipError("no HIP device visible: libpsmc_hip needs an AMD GPU (no CPU fallback)")
        self.n = int(n_states)
        self.mode = mode
        h = C.c_void_p()
        rc = self.lib.psmc_hip_create(C.byref(h), self.n, int(device), int(mode))
        if rc != 0:
            raise HipError("psmc_hip_create: %s" % self.lib.psmc_hip_strerror(rc).decode())
        self.h = h
        self.n_seg = 0
        self.n_sel = 0
        self._keep = None
        for k, v in options.items():
            self.set_option(k, v)

    def _chk(self, rc, what):
        if rc != 0:
            raise HipError("%s: %s (%s)" % (what, self.lib.psmc_hip_strerror(rc).decode(),
                                            self.lib.psmc_hip_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.psmc_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._chk(self.lib.psmc_hip_set_option(self.h, key.encode(), float(value)), "set_option(%s)" % key)

    def set_cu_range(self, first, count):
        """Mask the context's streams to `count` compute units from `first` (0: whole device); psmc_hip_set_cu_range."""
        self._chk(self.lib.psmc_hip_set_cu_range(self.h, int(first), int(count)), "set_cu_range")

    def reserve_tables(self):
        self._chk(self.lib.psmc_hip_reserve_tables(self.h), "reserve_tables")

    def reserve_batch_tables(self, max_bins):
        self.lib.psmc_hip_reserve_batch_tables.argtypes = [C.c_void_p, C.c_int64]
        self._chk(self.lib.psmc_hip_reserve_batch_tables(self.h, int(max_bins)), "reserve_batch_tables")

    def load_segments(self, segs):
        segs = [np.ascontiguousarray(s, dtype=np.uint8) for s in segs]
        n = len(segs)
        ptrs = (_u8p * n)(*[s.ctypes.data_as(_u8p) for s in segs])
        lens = np.array([len(s) for s in segs], dtype=np.int32)
        self._chk(self.lib.psmc_hip_load_segments(self.h, n, ptrs, lens.ctypes.data_as(_i32p)), "load_segments")
        self.n_seg = self.n_sel = n
        self.lens = lens

    def load_segments_device(self, d_obs_ptr, offsets, lens, keepalive=None):
        """Observations already in HBM (e.g. a torch.uint8 CUDA tensor's data_ptr())."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        self._keep = keepalive
        self._chk(self.lib.psmc_hip_load_segments_device(self.h, len(lens), C.c_void_p(int(d_obs_ptr)),
                                                         off.ctypes.data_as(_i64p), lens.ctypes.data_as(_i32p)),
                  "load_segments_device")
        self.n_seg = self.n_sel = len(lens)
        self.lens = lens

    def select(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._chk(self.lib.psmc_hip_select(self.h, len(idx), idx.ctypes.data_as(_i32p)), "select")
        self.n_sel = len(idx)

    def _params(self, a, e, a0):
        a = np.ascontiguousarray(a, dtype=np.float64)
        e = np.ascontiguousarray(np.asarray(e, dtype=np.float64)[:2])
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        assert a.shape == (self.n, self.n) and e.shape == (2, self.n) and a0.shape == (self.n,)
        return a, e, a0

    def estep(self, a, e, a0):
        """dict(A (n,n), E (2,n), A0 (n,), LL, chk (n_sel,)) -- em.c:33-55."""
        a, e, a0 = self._params(a, e, a0)
        n = self.n
        A = np.zeros((n, n)); E = np.zeros((2, n)); A0 = np.zeros(n); LL = C.c_double(0)
        chk = np.zeros(self.n_sel)
        self._chk(self.lib.psmc_hip_estep(self.h, _p(a), _p(e), _p(a0), _p(A), _p(E), _p(A0), C.byref(LL), _p(chk)),
                  "estep")
        return dict(A=A, E=E, A0=A0, LL=LL.value, chk=chk)

    def estep_segments(self, a, e, a0):
        """Exact mode: the per-segment `he` of em.c:49 (seg_A, seg_E incl. the missing row, seg_A0, seg_LL, chk)."""
        a, e, a0 = self._params(a, e, a0)
        n, ns = self.n, self.n_sel
        sA = np.zeros((ns, n, n)); sE = np.zeros((ns, 3, n)); sA0 = np.zeros((ns, n)); sLL = np.zeros(ns)
        chk = np.zeros(ns)
        self._chk(self.lib.psmc_hip_estep_segments(self.h, _p(a), _p(e), _p(a0), _p(sA), _p(sE), _p(sA0), _p(sLL),
                                                   _p(chk)), "estep_segments")
        return dict(seg_A=sA, seg_E=sE, seg_A0=sA0, seg_LL=sLL, chk=chk)

    def estep_batch(self, params, selections, want="A", on_done=None):
        """Config 4: one call for n_rep replicates.  params: list of (a, e, a0); selections: list of index lists
        (bootstrap multisets over the loaded segments).  want: "A" (full counts), "sums" (5 triangular sums) or "both".
        on_done(replicates, out): psmc_hip_estep_batch_cb's progress callback -- the positions whose rows of `out` are final now.
        -> dict(A (R,n,n) | sums (R,5,n), E (R,2,n), LL (R,))."""
        R, n = len(params), self.n
        assert len(selections) == R
        a = np.ascontiguousarray(np.stack([np.asarray(p[0], dtype=np.float64) for p in params]))
        e = np.ascontiguousarray(np.stack([np.asarray(p[1], dtype=np.float64)[:2] for p in params]))
        a0 = np.ascontiguousarray(np.stack([np.asarray(p[2], dtype=np.float64) for p in params]))
        assert a.shape == (R, n, n) and e.shape == (R, 2, n) and a0.shape == (R, n)
        off = np.concatenate([[0], np.cumsum([len(x) for x in selections])]).astype(np.int32)
        idx = np.concatenate([np.asarray(x, dtype=np.int32) for x in selections]).astype(np.int32)
        A = np.zeros((R, n, n)) if want in ("A", "both") else None
        sums = np.zeros((R, 5, n)) if want in ("sums", "both") else None
        E = np.zeros((R, 2, n)); LL = np.zeros(R)
        out = dict(E=E, LL=LL)
        if A is not None:
            out["A"] = A
        if sums is not None:
            out["sums"] = sums
        if on_done is None:
            self.lib.psmc_hip_estep_batch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _i32p, _i32p, _dp, _dp, _dp, _dp]
            self._chk(self.lib.psmc_hip_estep_batch(self.h, R, _p(a), _p(e), _p(a0), off.ctypes.data_as(_i32p), idx.ctypes.data_as(_i32p),
                                                    _p(A), _p(sums), _p(E), _p(LL)), "estep_batch")
            return out
        fn_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, _i32p)
        cb = fn_t(lambda user, n_done, reps: on_done([int(reps[i]) for i in range(n_done)], out))
        self.lib.psmc_hip_estep_batch_cb.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _i32p, _i32p, _dp, _dp, _dp, _dp, fn_t, C.c_void_p]
        self._chk(self.lib.psmc_hip_estep_batch_cb(self.h, R, _p(a), _p(e), _p(a0), off.ctypes.data_as(_i32p), idx.ctypes.data_as(_i32p),
                                                   _p(A), _p(sums), _p(E), _p(LL), cb, None), "estep_batch_cb")
        return out

    def batch_info(self):
        o = (C.c_int * 2)()
        self._chk(self.lib.psmc_hip_batch_info(self.h, o), "batch_info")
        return dict(groups=o[0], replicate_contexts=o[1])

    def estep_device(self, a, e, a0, d_stats_ptr, stream_ptr=0):
        """Fast mode, asynchronous: [A | E | LL] (n*n+2n+1 doubles) into device memory on `stream`."""
        a, e, a0 = self._params(a, e, a0)
        self._chk(self.lib.psmc_hip_estep_device(self.h, _p(a), _p(e), _p(a0), C.c_void_p(int(d_stats_ptr)),
                                                 C.c_void_p(int(stream_ptr))), "estep_device")

    def estep_factored(self, a, e, a0):
        """Fast mode, PSMC-form matrix: dict(sums (5, n) = SL, SU, DG, CL, CU; E (2, n); LL) without the N x N counts."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        e = np.ascontiguousarray(np.asarray(e, dtype=np.float64)[:2])
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        n = self.n
        sums = np.zeros((5, n)); E = np.zeros((2, n)); LL = C.c_double(0)
        self.lib.psmc_hip_estep_factored.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_double)]
        self._chk(self.lib.psmc_hip_estep_factored(self.h, _p(a), _p(e), _p(a0), _p(sums), _p(E), C.byref(LL)), "estep_factored")
        return dict(sums=sums, E=E, LL=LL.value)

    def fast_diag(self):
        wf = C.c_double(0); wb = C.c_double(0); nc = C.c_int(0); wu = C.c_int(0)
        self._chk(self.lib.psmc_hip_fast_diag(self.h, C.byref(wf), C.byref(wb), C.byref(nc), C.byref(wu)), "fast_diag")
        rp = (C.c_int * 6)()
        self._chk(self.lib.psmc_hip_fast_repairs(self.h, rp), "fast_repairs")
        fi = (C.c_int * 8)()
        self._chk(self.lib.psmc_hip_fast_info(self.h, fi), "fast_info")
        return dict(warm_err_fwd=wf.value, warm_err_bwd=wb.value, n_chunks=nc.value, warmup=wu.value,
                    fwd_rounds=rp[0], bwd_rounds=rp[1], fwd_tiles=rp[2], bwd_tiles=rp[3], merged=rp[4], recounted=rp[5],
                    structured=bool(fi[0]), tile_len=fi[1], items_fwd=fi[2], items_bwd=fi[3], back_half=fi[4], ckpt=bool(fi[5]),
                    fused_launches=fi[6], merged_phase1=fi[7])

    def fast_plan(self):
        """The plan of the next fast E-step: tiles, tile length, mean / longest warm-ups (bins), glued tiles."""
        o = np.zeros(8)
        self.lib.psmc_hip_fast_plan.argtypes = [C.c_void_p, _dp]
        self._chk(self.lib.psmc_hip_fast_plan(self.h, _p(o)), "fast_plan")
        return dict(tiles=int(o[0]), tile_len=int(o[1]), warm_fwd_mean=o[2], warm_bwd_mean=o[3], warm_fwd_max=int(o[4]), warm_bwd_max=int(o[5]),
                    glued_fwd=int(o[6]), glued_bwd=int(o[7]))

    def tables(self, seg, want_b=True):
        L = int(self.lens[seg])
        f = np.zeros((L, self.n)); s = np.zeros(L)
        b = np.zeros((L, self.n)) if want_b else None
        self._chk(self.lib.psmc_hip_get_tables(self.h, int(seg), _p(f), _p(b), _p(s)), "get_tables")
        return f, b, s

    def decode(self, seg):
        """(path, maxp): posterior-argmax state and its probability per bin (khmm.c:264-281), exact mode."""
        L = int(self.lens[seg])
        path = np.zeros(L, dtype=np.int32); mp = np.zeros(L)
        self._chk(self.lib.psmc_hip_decode(self.h, int(seg), path.ctypes.data_as(_i32p), _p(mp)), "decode")
        return path, mp

    def posterior(self, seg, want_post=True, want_recomb=True):
        """(post (L, n), recomb (L,)): full posterior and the DF line's recombination probability (aux.c:183-200), exact mode."""
        L = int(self.lens[seg])
        post = np.zeros((L, self.n)) if want_post else None
        rec = np.zeros(L) if want_recomb else None
        self.lib.psmc_hip_posterior.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        self._chk(self.lib.psmc_hip_posterior(self.h, int(seg), _p(post), _p(rec)), "posterior")
        return post, rec

    def post_counts(self, seg, cnt1, cnt):
        """cnt (n, n_cnt) += posterior-weighted counts of segment `seg` (cnt1: (l, n_cnt) int32), aux.c:202-219; in place."""
        cnt1 = np.ascontiguousarray(cnt1, dtype=np.int32)
        assert cnt.dtype == np.float64 and cnt.flags.c_contiguous and cnt.shape == (self.n, cnt1.shape[1])
        self.lib.psmc_hip_post_counts.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int32, C.c_int32, _dp]
        self._chk(self.lib.psmc_hip_post_counts(self.h, int(seg), cnt1.ctypes.data_as(_i32p), cnt1.shape[0], cnt1.shape[1], _p(cnt)), "post_counts")
        return cnt

    def timing(self):
        ms = np.zeros(7)
        self._chk(load_diag().psmc_hip_last_timing(self.h, _p(ms)), "last_timing")
        if self.mode == MODE_FAST:  # forward and backward chains run concurrently: see include/psmc_hip.h
            return dict(total=ms[0], chains=ms[1], tail=ms[2], expect=ms[3], reduce=ms[4], fwd_sweep=ms[5],
                        bwd_sweep=ms[6], forward=ms[1], backward=ms[2])
        return dict(total=ms[0], forward=ms[1], backward=ms[2], expect=ms[3], reduce=ms[4], fwd_sweep=0.0, bwd_sweep=0.0)


def selftest(device=0):
    lib = load_diag()
    return lib.psmc_hip_selftest(int(device))


MICROBENCH_NAMES = ["fmac_dpp dependent", "v_fma_f64 dependent", "v_add_f64 dependent", "fmac_dpp 4 chains (per op)",
                    "fmac_dpp 8 chains (per op)", "rep_rows_swap dependent", "rep_rows_bperm dependent",
                    "dpp_mov+add dependent", "v_rcp_f64 dependent", "mov_b64_dpp+mul dependent",
                    "v_mul_f64 8 indep (per op)", "f64 division dependent", "mfma_f64_16x16x4 dependent",
                    "mfma_f64_16x16x4 4 accumulators (per op)", "structured step dependent (per step)",
                    "structured step + emission + norm/4 (per step)", "cycle counter MHz (vs 100 MHz wall clock)",
                    "one-state-per-lane structured step + emission + norm/4 (per step)",
                    "4 mfma_f64 + 32 independent v_fma_f64 (per group; 4 mfma alone: 4x the 4-accumulator figure)",
                    "4 mfma_f64 + 32 v_mov_b32_dpp (per group)"]


def microbench(device=0):
    lib = load_diag()
    out = np.zeros(len(MICROBENCH_NAMES))
    lib.psmc_hip_microbench.argtypes = [C.c_int, _dp, C.c_int]
    rc = lib.psmc_hip_microbench(int(device), _p(out), len(MICROBENCH_NAMES))
    if rc != 0:
        raise HipError("microbench: %s" % load_library().psmc_hip_strerror(rc).decode())
    return dict(zip(MICROBENCH_NAMES, out.tolist()))


def hbm_probe(nbytes=8 << 30, device=0):
    """Achievable HBM rates of plain streaming kernels (GB/s): fill, read, copy, sweep-like stores."""
    lib = load_diag()
    lib.psmc_hip_hbm_probe.argtypes = [C.c_int, C.c_longlong, _dp]
    out = np.zeros(4)
    rc = lib.psmc_hip_hbm_probe(int(device), int(nbytes), _p(out))
    if rc != 0:
        raise HipError("hbm_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    return dict(zip(["fill", "read", "copy", "sweep_store"], out.tolist()))


def load_probe(n_waves, steps=20000, device=0):
    """The structured step on n_waves waves at once: kernel ms, mean / max cycles per step, mean shader MHz."""
    lib = load_diag()
    lib.psmc_hip_load_probe.argtypes = [C.c_int, C.c_int, C.c_int, _dp]
    out = np.zeros(4)
    rc = lib.psmc_hip_load_probe(int(device), int(n_waves), int(steps), _p(out))
    if rc != 0:
        raise HipError("load_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    return dict(zip(["ms", "cycles_per_step", "max_cycles_per_step", "mhz"], out.tolist()))


def stream_probe(n_doubles=1 << 27, device=0):
    """Known-size 8 B/lane copy (for PMC calibration); returns (ms per launch, GB/s read+write)."""
    lib = load_diag()
    lib.psmc_hip_stream_probe.argtypes = [C.c_int, C.c_longlong, _dp]
    ms = C.c_double(0)
    rc = lib.psmc_hip_stream_probe(int(device), int(n_doubles), C.byref(ms))
    if rc != 0:
        raise HipError("stream_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    return ms.value, 16.0 * n_doubles / (ms.value * 1e-3) / 1e9


PIPE_PROBE_CONFIGS = ["4 matrix waves (1/SIMD)", "4 vector waves (1/SIMD)", "8 matrix waves (2/SIMD)", "8 vector waves (2/SIMD)",
                      "waves 0-3 matrix + 4-7 vector (one of each per SIMD)", "even waves matrix, odd vector (SIMDs not mixed)"]


def pipe_probe(device=0):
    """Cross-wave overlap of f64 matrix and f64 vector instructions on one SIMD: {configuration: cycles per round of each wave}."""
    lib = load_diag()
    lib.psmc_hip_pipe_probe.argtypes = [C.c_int, _dp, C.c_int]
    out = np.zeros(8 * len(PIPE_PROBE_CONFIGS))
    rc = lib.psmc_hip_pipe_probe(int(device), _p(out), len(out))
    if rc != 0:
        raise HipError("pipe_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    return {name: [round(v, 1) for v in out[8 * i:8 * i + 8] if v > 0] for i, name in enumerate(PIPE_PROBE_CONFIGS)}


PIPE_KINDS = {"idle": 0, "mfma_f64": 1, "fma_f64": 2, "mov_dpp": 3, "scan_levels": 4, "ds_read_b128": 5, "sload_readlane": 6,
              "add_u32": 7, "fma_f32": 8, "add_f64": 9}


def pipe_probe2(kinds, rounds=64, device=0):
    """Cycles per round of up to 8 waves of one work-group (wave w -> SIMD w % 4), kinds[w] from PIPE_KINDS."""
    lib = load_diag()
    k = (C.c_int * 8)(*([PIPE_KINDS[x] if isinstance(x, str) else int(x) for x in kinds] + [0] * (8 - len(kinds))))
    out = np.zeros(8)
    lib.psmc_hip_pipe_probe2.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, _dp]
    rc = lib.psmc_hip_pipe_probe2(int(device), k, int(rounds), _p(out))
    if rc != 0:
        raise HipError("pipe_probe2: %s" % load_library().psmc_hip_strerror(rc).decode())
    return [float(v) for v in out[:len(kinds)]]


def place_probe(n_waves, waves_per_block=1, n_kernels=1, steps=3328, device=0):
    """Where the waves of small launches land: dict(ms, cycles_per_step mean/max, simds_used, max_waves_per_simd, hist)."""
    lib = load_diag()
    npad = (n_waves + waves_per_block - 1) // waves_per_block * waves_per_block
    out = np.zeros(3 * npad * n_kernels); ms = C.c_double(0)
    lib.psmc_hip_place_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_double)]
    rc = lib.psmc_hip_place_probe(int(device), int(n_waves), int(waves_per_block), int(n_kernels), int(steps), _p(out), C.byref(ms))
    if rc != 0:
        raise HipError("place_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    o = out.reshape(-1, 3)
    o = o[o[:, 0] > 0]
    hw = o[:, 1].astype(np.int64); xcc = o[:, 2].astype(np.int64) & 0xf
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    _, cnt = np.unique(key, return_counts=True)
    cus = len(np.unique(key // 4))
    return dict(ms=ms.value, cycles_mean=float(o[:, 0].mean()), cycles_max=float(o[:, 0].max()), waves=len(o), simds_used=len(cnt), cus_used=cus,
                max_waves_per_simd=int(cnt.max()), hist={int(k): int((cnt == k).sum()) for k in np.unique(cnt)})


def cumask_probe(n_cus_a, n_waves_a, n_waves_b, steps=3328, device=0):
    """Two concurrent launches on streams with complementary compute-unit masks (stream A: the first n_cus_a units of the
    mask's bit order; 0 = no masks): where their waves land.  dict(ms, a=..., b=..., shared_cus) with per-launch cus_used,
    simds_used, max_waves_per_simd, cycles_mean, and the number of compute units both touched."""
    lib = load_diag()
    out = np.zeros(3 * (n_waves_a + n_waves_b)); ms = C.c_double(0)
    lib.psmc_hip_cumask_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_double)]
    rc = lib.psmc_hip_cumask_probe(int(device), int(n_cus_a), int(n_waves_a), int(n_waves_b), int(steps), _p(out), C.byref(ms))
    if rc != 0:
        raise HipError("cumask_probe: %s" % load_library().psmc_hip_strerror(rc).decode())
    o = out.reshape(-1, 3)

    def summary(q):
        hw = q[:, 1].astype(np.int64); xcc = q[:, 2].astype(np.int64) & 0xf
        simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        _, cnt = np.unique(cukey * 4 + simd, return_counts=True)
        per_xcc = {int(x): int(len(np.unique(cukey[xcc == x]))) for x in np.unique(xcc)}
        return dict(waves=len(q), cus_used=len(np.unique(cukey)), simds_used=len(cnt), max_waves_per_simd=int(cnt.max()),
                    cycles_mean=float(q[:, 0].mean()), cycles_max=float(q[:, 0].max()), cus_per_xcc=per_xcc), set(cukey.tolist())
    a, ka = summary(o[:n_waves_a]); b, kb = summary(o[n_waves_a:])
    # the compute units stream A touched, as (xcc, se, sh, cu): which units the first n_cus_a bits of the mask name
    a["cus"] = sorted((k >> 8, (k >> 5) & 7, (k >> 4) & 1, k & 15) for k in ka)
    return dict(ms=ms.value, a=a, b=b, shared_cus=len(ka & kb))


class HipGroup:
    """One E-step sharded over several GPUs inside the C library (psmc_hip_group_*): LPT partition of the segments,
    per-device E-steps on host threads, RCCL all-reduce of [A | E | LL] (fast) or ordered per-segment sum (exact)."""

    def __init__(self, n_states, devices, mode=MODE_FAST, **options):
        self.lib = load_library()
        if self.lib.psmc_hip_device_count() <= 0:
            raise HipError("no HIP device visible: libpsmc_hip needs an AMD GPU (no CPU fallback)")
        self.n = int(n_states); self.mode = mode
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        self.lib.psmc_hip_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        rc = self.lib.psmc_hip_group_create(C.byref(g), self.n, len(devices), devs, int(mode))
        if rc != 0:
            raise HipError("psmc_hip_group_create: %s" % self.lib.psmc_hip_strerror(rc).decode())
        self.g = g
        self.lib.psmc_hip_group_last_error.restype = C.c_char_p
        self.lib.psmc_hip_group_last_error.argtypes = [C.c_void_p]
        self.lib.psmc_hip_group_destroy.argtypes = [C.c_void_p]
        self.lib.psmc_hip_group_destroy.restype = None
        self.lib.psmc_hip_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        self.lib.psmc_hip_group_load_segments.argtypes = [C.c_void_p, C.c_int, C.POINTER(_u8p), _i32p]
        self.lib.psmc_hip_group_estep.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        self.lib.psmc_hip_group_estep_factored.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_double)]
        self.lib.psmc_hip_group_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), _i32p, C.POINTER(C.c_int)]
        self.n_seg = 0
        for k, v in options.items():
            self.set_option(k, v)

    def _chk(self, rc, what):
        if rc != 0:
            raise HipError("%s: %s (%s)" % (what, self.lib.psmc_hip_strerror(rc).decode(), self.lib.psmc_hip_group_last_error(self.g).decode()))

    def close(self):
        if getattr(self, "g", None):
            self.lib.psmc_hip_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._chk(self.lib.psmc_hip_group_set_option(self.g, key.encode(), float(value)), "group_set_option(%s)" % key)

    def load_segments(self, segs):
        segs = [np.ascontiguousarray(s, dtype=np.uint8) for s in segs]
        n = len(segs)
        ptrs = (_u8p * n)(*[s.ctypes.data_as(_u8p) for s in segs])
        lens = np.array([len(s) for s in segs], dtype=np.int32)
        self._chk(self.lib.psmc_hip_group_load_segments(self.g, n, ptrs, lens.ctypes.data_as(_i32p)), "group_load_segments")
        self.n_seg = n

    def estep(self, a, e, a0):
        a = np.ascontiguousarray(a, dtype=np.float64); e = np.ascontiguousarray(np.asarray(e, dtype=np.float64)[:2])
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        n = self.n
        A = np.zeros((n, n)); E = np.zeros((2, n)); A0 = np.zeros(n); LL = C.c_double(0); chk = np.zeros(self.n_seg)
        self._chk(self.lib.psmc_hip_group_estep(self.g, _p(a), _p(e), _p(a0), _p(A), _p(E), _p(A0), C.byref(LL), _p(chk)), "group_estep")
        return dict(A=A, E=E, A0=A0, LL=LL.value, chk=chk)

    def estep_factored(self, a, e, a0):
        a = np.ascontiguousarray(a, dtype=np.float64); e = np.ascontiguousarray(np.asarray(e, dtype=np.float64)[:2])
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        n = self.n
        sums = np.zeros((5, n)); E = np.zeros((2, n)); LL = C.c_double(0)
        self._chk(self.lib.psmc_hip_group_estep_factored(self.g, _p(a), _p(e), _p(a0), _p(sums), _p(E), C.byref(LL)), "group_estep_factored")
        return dict(sums=sums, E=E, LL=LL.value)

    def selfcheck(self):
        """First contact with the devices of the group: dict(shards, path ('rccl' / 'host_sum' / 'single' / 'exact'), communicator,
        note) -- raises HipError naming the step that failed (psmc_hip_group_selfcheck)."""
        out = (C.c_int * 4)()
        self.lib.psmc_hip_group_selfcheck.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._chk(self.lib.psmc_hip_group_selfcheck(self.g, out), "group_selfcheck")
        note = self.lib.psmc_hip_group_last_error(self.g).decode() if out[3] else ""
        return dict(shards=out[0], path={0: "single", 1: "rccl", 2: "host_sum", 3: "exact"}[out[1]], communicator=bool(out[2]), note=note)

    def info(self):
        ns = C.c_int(0); lr = C.c_int(0); so = np.zeros(max(self.n_seg, 1), dtype=np.int32)
        self._chk(self.lib.psmc_hip_group_info(self.g, C.byref(ns), so.ctypes.data_as(_i32p), C.byref(lr)), "group_info")
        return dict(n_shards=ns.value, shard_of_seg=so[:self.n_seg].tolist(), last_reduce=lr.value)
