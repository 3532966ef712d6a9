import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    """True when the C-ABI library is built and sees a HIP device (asked through the library itself, no torch)."""
    try:
        from psmc_amd import hip
        return hip.load_library().psmc_hip_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a GPU: gpu-marked tests are skipped, not failed.  An explicit `-m gpu`
    run is left alone -- on the GPU box a missing device or library must FAIL loudly, not skip."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu-marked test)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _npz(name):
    return dict(np.load(os.path.join(GOLD, name)))


class Golden:
    """Golden vectors dumped from the real reference by tests/golden/make_golden.py."""

    def __init__(self):
        self.params_raw = _npz("hmm_params.npz")
        self.small = _npz("estep_small.npz")
        self.mid = _npz("estep_mid.npz")
        self.kats = _npz("host_kats.npz")
        self.n128 = _npz("estep_n128.npz")  # make_golden_n128.py: -p "64*2", 128 states
        s = _npz("segments_small.npz")
        self.segs_small = [s[k] for k in sorted(s)]
        s = _npz("segments_mid.npz")
        self.segs_mid = [s[k] for k in sorted(s)]

    def params(self, key):
        """dict(a, e (3,n), a0, sigma, t, params, pattern, ...) of one parameter set, e.g. 'n64_curve'."""
        pre = key + "."
        d = {k[len(pre):]: v for k, v in self.params_raw.items() if k.startswith(pre)}
        d["pattern"] = str(d["pattern"])
        return d

    def param_keys(self):
        return sorted({k.split(".")[0] for k in self.params_raw})


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.build_oracle()
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    import orc
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference checkout not present (GPU box)")
    orc.build_oracle(with_ref=True)
    return orc.Reference()


FAST_METRICS = []   # psmc_amd.parity.fast_error_metrics of every fast-vs-reference comparison of the session (check_fast)


def pytest_terminal_summary(terminalreporter):
    if not FAST_METRICS:
        return
    keys = sorted({k for m in FAST_METRICS for k in m})
    worst = {k: max(m[k] for m in FAST_METRICS if k in m) for k in keys}
    terminalreporter.write_line("fast mode vs exact / oracle, worst of %d comparisons: %s" % (
        len(FAST_METRICS), "  ".join("%s %.2e" % (k, worst[k]) for k in keys)))


def bits_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    return x.shape == y.shape and np.array_equal(x.view(np.uint64), y.view(np.uint64))
