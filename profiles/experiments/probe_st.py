import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np
from psmc_amd import hip
lib = hip.load_library()
lib.psmc_hip_load_probe_st.argtypes = [C.c_int]*5 + [C.POINTER(C.c_double)]
def run(nw, steps, st, mode):
    out = np.zeros(5)
    rc = lib.psmc_hip_load_probe_st(0, nw, steps, st, mode, out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, dict(ms=round(out[0],3), cyc=round(out[1]), cmax=round(out[2]), mhz=round(out[3]), gbs=round(out[4]))
for nw in (1024, 1920, 2048):
    for steps, st in ((7808, 3712), (3712, 3712)):
        for mode in (1, 2):
            print(nw, steps, st, "mode", mode, run(nw, steps, st, mode))
    print(nw, "no stores", hip.load_probe(nw, 7808))
