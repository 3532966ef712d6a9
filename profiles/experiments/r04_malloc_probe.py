#!/usr/bin/env python3
"""How long does hipMalloc take for table-sized allocations (the exact batch holds ~250 GB of f/b tables)?  (round 4 probe)"""
import ctypes as C, time, sys
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r
assert hip.hipSetDevice(0) == 0
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for gb in (1, 8, 32, 128, 128, 250):
    n = gb << 30
    dt, rc = t(lambda: hip.hipMalloc(C.byref(p), n))
    if rc != 0: print("hipMalloc %d GB: rc %d" % (gb, rc)); continue
    d1, _ = t(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
    d2, _ = t(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
    d3, _ = t(lambda: hip.hipFree(p))
    print("%4d GB: hipMalloc %.3f s, first memset %.3f s, second memset %.3f s, hipFree %.3f s" % (gb, dt, d1, d2, d3), flush=True)
# part 2: is it the size of one allocation, or the total?  30 x 8 GB, then 15 x 16 GB
for gb, cnt in ((8, 30), (16, 15), (24, 10)):
    ps, ts = [], []
    for i in range(cnt):
        q = C.c_void_p()
        dt, rc = t(lambda: hip.hipMalloc(C.byref(q), gb << 30))
        if rc != 0: print("  alloc %d of %d GB failed rc %d" % (i, gb, rc)); break
        ps.append(q); ts.append(dt)
    d1, _ = t(lambda: ([hip.hipMemset(q, 0, gb << 30) for q in ps], hip.hipDeviceSynchronize()))
    print("%d x %d GB: hipMalloc total %.3f s (max %.3f), memset of all %.3f s" % (len(ps), gb, sum(ts), max(ts), d1), flush=True)
    d3, _ = t(lambda: [hip.hipFree(q) for q in ps])
    print("   free %.3f s" % d3)
