"""psmc_amd -- MI355X-native E-step (Baum-Welch forward-backward + expected
counts) of lh3/psmc.  The compute lives in libpsmc_hip.so (hand-written gfx950
HIP kernels behind the C-ABI of include/psmc_hip.h); this package is only the
thin ctypes mirror used by tests, bench.py and tooling."""
from .hip import HipEStep, HipError, load_library, lib_path  # noqa: F401

__all__ = ["HipEStep", "HipError", "load_library", "lib_path"]
