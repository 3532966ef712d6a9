#!/usr/bin/env python3
"""Per-wave timeline of phase 1 of a fast-mode E-step (debug build: make -C psmc_amd/csrc EXTRA=-DPSMC_TRACE_SWEEP).

Every wave of the bulk forward sweep and of the backward warm-up pass stamps its start, the first block it stores (the end of
its warm-up) and its end with the 100 MHz wall clock, plus the SIMD it ran on.  Prints percentiles of those stamps relative to
the earliest start, and the same split by "first / second wave on its SIMD".

    python scripts/sweep_trace.py [share_N] ["opt=v opt=v"]     # share_N: rank 0's share of the genome at N GPUs (1 = whole genome, 0 = one 500 k-bin chromosome)
"""
import ctypes as C
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    opts = sys.argv[2].split() if len(sys.argv) > 2 else []
    import torch
    import bench
    from psmc_amd import hip, sim
    from psmc_amd.dist import partition_segments
    a, e, a0 = bench.load_params()
    traj, _ = bench.load_trajectory(os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
    lens = sim.human_like_lengths(30_000_000, n_seg=90)
    full = sim.simulate_genome(a, e, a0, lens, seed=43)
    if share == 0:   # config 2: one 500 k-bin chromosome
        segs = [sim.simulate_segment(a, e, a0, 500000, np.random.default_rng(7))]
    else:
        segs = [full[i] for i in partition_segments(lens, share)[0]]
    sh = bench.Shard(hip, torch, segs, 64, 0, hip.MODE_FAST, opts)
    es = sh.es
    stream = torch.cuda.current_stream()
    es.estep(*traj[0])
    for i in range(int(os.environ.get("TRACE_WARM", "10"))):   # (adaptive warm-ups, "adapt=1", settle over ~30 steps)
        es.estep_device(*traj[i % len(traj)], sh.stats.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    d = es.fast_diag()
    lib = hip.load_library()
    # round 4: one line per step over a few more steps -- total, and how the walks were placed (two walks on one SIMD run at half speed)
    for i in range(int(os.environ.get("TRACE_STEPS", "0"))):
        es.estep_device(*traj[(10 + i) % len(traj)], sh.stats.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        buf = np.zeros(4 * 16384, dtype=np.uint64)
        assert lib.psmc_hip_debug_trace(2, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 16384) == 0
        t = buf.reshape(16384, 4); t = t[t[:, 0] > 0]; t = t[t[:, 0] >= t[:, 0].max() - np.uint64(2_000_000)]
        hw = t[:, 3]; h = hw.astype(np.int64) & 0xFFFFFFFF; xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
        simd_id = ((((xcc * 8 + ((h >> 13) & 7)) * 2 + ((h >> 12) & 1)) * 16 + ((h >> 8) & 0xF)) * 4 + ((h >> 4) & 3))
        dur = (t[:, 2].astype(np.int64) - t[:, 0].astype(np.int64)) / 100.0
        cnt = np.bincount(np.unique(simd_id, return_counts=True)[1])
        k = es.timing()
        print("step %2d total %.3f ms | walks %d on %d SIMDs, SIMDs holding 1/2/3+ walks: %s | walk us p50 %.0f p90 %.0f max %.0f | start spread %.0f us" % (
            i, k["total"], len(t), len(set(simd_id.tolist())), cnt[1:].tolist(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max(), (t[:, 0].max() - t[:, 0].min()) / 100.0), flush=True)
    nw = (d["items_fwd"] + 3) // 4
    print("tiles %d x %d, fwd items %d, bwd items %d, kernels %s" % (d["n_chunks"], d["tile_len"], d["items_fwd"], d["items_bwd"], {k: round(v, 2) for k, v in es.timing().items()}))
    walk_simds, walk_cus = {}, set()
    for which, name in ((2, "walks"), (0, "forward bulk sweep"), (1, "backward warm-up pass")):
        n = 16384
        buf = np.zeros(4 * n, dtype=np.uint64)
        rc = lib.psmc_hip_debug_trace(which, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), n)
        assert rc == 0, rc
        t = buf.reshape(n, 4)
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] >= t[:, 0].max() - np.uint64(2_000_000)]   # stamps of the last launch only (20 ms window)
        if not len(t):
            print(name, ": no stamps"); continue
        if which == 2:
            t0w = t[:, 0].min()
        t0 = t[:, 0].min()
        us = lambda c: (c.astype(np.int64) - np.int64(t0)) / 100.0
        st, wu, en = us(t[:, 0]), us(t[:, 1]), us(t[:, 2])
        hw = t[:, 3]
        steps = (hw >> np.uint64(40)).astype(np.int64) * 16   # steps of the wave's longest row
        h = hw.astype(np.int64) & 0xFFFFFFFF; xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF   # as psmc_amd/hip.py place_probe
        simd_id = ((((xcc * 8 + ((h >> 13) & 7)) * 2 + ((h >> 12) & 1)) * 16 + ((h >> 8) & 0xF)) * 4 + ((h >> 4) & 3))
        order = np.argsort(st, kind="stable")
        seen, rank_on_simd = {}, np.zeros(len(t), dtype=int)
        for i in order:
            rank_on_simd[i] = seen.get(simd_id[i], 0); seen[simd_id[i]] = rank_on_simd[i] + 1
        cus = set((simd_id >> 2).tolist())
        if which == 2:
            walk_cus = cus
        by_xcc = np.bincount(xcc, minlength=8).tolist()
        by_se = np.bincount(((h >> 13) & 7), minlength=4).tolist()
        cu_in_se = sorted(set(((h >> 8) & 0xF).tolist()))
        pc = lambda x: "min %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
        print("%s: %d waves on %d SIMDs of %d compute units (%d of them also run walks; waves per SIMD: %s)" % (
            name, len(t), len(seen), len(cus), len(cus & walk_cus), dict(zip(*np.unique(list(seen.values()), return_counts=True)))))
        print("   waves per XCC %s, per SE id %s, CU ids in use %s" % (by_xcc, by_se, cu_in_se))
        print("   start        us: " + pc(st))
        if which == 0:
            ok = t[:, 1] > 0
            print("   warm-up done us: " + pc(wu[ok]))
        print("   end          us: " + pc(en))
        if which == 1:
            mhz = t[:, 1].astype(np.float64) / np.maximum(en - st, 1e-3)
            print("   shader clock over the wave's life (s_memtime cycles / wall-clock us), MHz: " + pc(mhz))
        for r in range(int(rank_on_simd.max()) + 1):
            sel = rank_on_simd == r
            if which == 0:
                print("   wave #%d on its SIMD (%4d): warm-up done p50 %7.0f  end p50 %7.0f  p90 %7.0f" % (r, sel.sum(), np.percentile(wu[sel & (t[:, 1] > 0)], 50), np.percentile(en[sel], 50), np.percentile(en[sel], 90)))
            else:
                print("   wave #%d on its SIMD (%4d): end p50 %7.0f  p90 %7.0f" % (r, sel.sum(), np.percentile(en[sel], 50), np.percentile(en[sel], 90)))
        if which == 2:
            for sid, a_, b_ in zip(simd_id, st, en):
                walk_simds[sid] = max(walk_simds.get(sid, 0.0), b_ - a_)
            print("   walk duration us: " + pc(en - st))
            continue
        shared = np.array([sid in walk_simds for sid in simd_id])
        print("   waves on a SIMD that also ran a walk: %d, end us p50 %7.0f p90 %7.0f max %7.0f | the others: end us p50 %7.0f p90 %7.0f p99 %7.0f max %7.0f" % (
            shared.sum(), *(np.percentile(en[shared], q) if shared.sum() else 0 for q in (50, 90, 100)), *(np.percentile(en[~shared], q) for q in (50, 90, 99, 100))))
        typical = np.median(steps)
        for lab, sel in (("typical rows (<= 1.1 x median %d steps)" % typical, steps <= 1.1 * typical), ("long rows (learned warm-ups)", steps > 1.1 * typical)):
            if sel.sum():
                print("   %-44s %4d waves: steps p50 %6d max %6d | end us p50 %7.0f  p90 %7.0f  p99 %7.0f  max %7.0f | us per step p50 %.3f" % (
                    lab, sel.sum(), np.percentile(steps[sel], 50), steps[sel].max(), np.percentile(en[sel], 50), np.percentile(en[sel], 90), np.percentile(en[sel], 99), en[sel].max(),
                    np.percentile((en[sel] - st[sel]) / np.maximum(steps[sel], 1), 50)))
    # the fused back half: shader clock under the matrix instructions
    buf = np.zeros(4 * 8192, dtype=np.uint64)
    if lib.psmc_hip_debug_trace_counts(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 8192) == 0:
        t = buf.reshape(8192, 4); t = t[t[:, 0] > 0]
        t = t[t[:, 0] >= t[:, 0].max() - np.uint64(2_000_000)]
        if len(t):
            dur = (t[:, 2].astype(np.int64) - t[:, 0].astype(np.int64)) / 100.0
            mhz = t[:, 1].astype(np.float64) / np.maximum(dur, 1e-3)
            q = lambda x: "min %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
            print("fused backward + counts (k_bwd_count4f_struct): %d waves; wave duration us: %s" % (len(t), q(dur)))
            print("   shader clock over the wave's life, MHz: " + q(mhz))
    sh.close()


if __name__ == "__main__":
    main()
