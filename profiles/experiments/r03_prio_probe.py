#!/usr/bin/env python3
"""Does s_setprio decide who issues?  2048 one-wave work-groups of the bare structured step (two per SIMD): plain, and with the first
half of the grid at wave priority 0 and the second (the younger wave of every SIMD) at 3 (PSMC_HIP_PROBE_PRIO=1).  Cycles per step of either half."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from psmc_amd import hip
lib = hip.load_library()
n = 2048
out = np.zeros(3 * n); ms = C.c_double(0)
lib.psmc_hip_place_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
rc = lib.psmc_hip_place_probe(0, n, 1, 1, 3328, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ms))
assert rc == 0, rc
o = out.reshape(-1, 3)
print("PSMC_HIP_PROBE_PRIO=%s: %.3f ms; cycles per step, first half of the grid: mean %.0f (min %.0f max %.0f); second half: mean %.0f (min %.0f max %.0f)" % (
    os.environ.get("PSMC_HIP_PROBE_PRIO"), ms.value, o[:n // 2, 0].mean(), o[:n // 2, 0].min(), o[:n // 2, 0].max(), o[n // 2:, 0].mean(), o[n // 2:, 0].min(), o[n // 2:, 0].max()))
