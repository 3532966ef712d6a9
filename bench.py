#!/usr/bin/env python3
"""bench.py -- PSMC E-step throughput on MI355X.

Metric (BASELINE.json): genome bins/sec through forward-backward at n=64 states.
A "step" is one E-step pass (forward sweep, backward sweep, expected counts,
reduction, and for N>1 the all-reduce of the sufficient statistics) over one
synthetic whole-genome batch (config 3: ~30 M bins, 90 segments shaped like the
human autosomes + scaffolds, longest 2.49e6 bins, drawn from a 64-state PSMC
model).  Observations are resident in HBM before the timed region; per step only
the 33 KB of HMM parameters cross PCIe, exactly as in an EM iteration.

THE PARAMETERS MOVE: every step gets the next parameter set of a trajectory taken
from a real EM run (tests/golden/traj_n64.json: PA lines of `psmc -N25` on this
workload, mapped to (a, e, a0) by the host library), cycling through the 25 rounds --
as in an EM run, the tile plan a context learned meets new parameters every
iteration.  The headline `value` is measured that way; `steady_state` (the same
parameters every step: the best case, what round 1 of this repo reported) is
given beside it.

    python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Scaling.  Segments are independent given the parameters (em.c:36-55); the one
exchange per step is the RCCL all-reduce of n*n+2n+1 doubles that replaces
hmm_add_expect (khmm.c:346).  Default "weak": every rank holds its own
genome-sized shard.  `--scaling strong` is config 3 proper: ONE 30 M-bin genome,
its segments spread over the ranks by longest-processing-time-first; with N>1 the
default run also reports that as `strong_scaling` beside the headline.
"""
import argparse
import json
import os
import sys
import time

# the fast E-step runs five kernels side by side; HIP's default of 4 hardware queues per process makes two
# of its streams share one (must be set before the HIP runtime starts, i.e. before torch is imported)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_STATES = 64
PATTERN = "4+25*2+4+6"               # README:12 of the reference: 64 states, 28 free lambdas
BYTES_PER_BIN = 16 * N_STATES + 18   # obs x2, f write+read, s write+read
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec (psmc_hip_hbm_probe: 5.3-5.9 TB/s streaming on this box)
F64_PEAK_TFLOPS = 78.6               # dense FP64, vector or v_mfma_f64_16x16x4 (they share the pipe: psmc_hip_microbench)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_params():
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_params.npz"))
    return g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]


def load_trajectory(path, n_sets=25):
    """Parameter sets of consecutive EM rounds (rounds 1..n_sets of the run the file records) -> [(a, e, a0)]."""
    from psmc_amd import hostlib
    tj = json.load(open(path))
    assert tj["pattern"] == PATTERN
    rounds = [r for r in tj["rounds"] if r["round"] >= 1][:n_sets]
    return [hostlib.hmm_params(PATTERN, r["params"]) for r in rounds], tj.get("source", path)


def cpu_baseline(a, e, a0, segs, sample_bins):
    """Time the CPU E-step on a bounded sample of the same workload (rank 0, N=1 only):
    the reference itself (oracle/_ref, built from its own sources) when that .so travelled
    with the repo, else our restatement of it."""
    import orc
    sample, tot = [], 0
    for s in segs:                       # 500k-bin trunks like utils/splitfa.c:35 of the reference
        for j in range(0, len(s), 500000):
            if tot >= sample_bins:
                break
            t = s[j:j + min(500000, sample_bins - tot)]
            sample.append(t); tot += len(t)
    if orc.have_reference():
        eng, kind = orc.Reference(), "reference"
    else:
        eng, kind = orc.Oracle(), "port"
    t0 = time.perf_counter()
    eng.estep(a, e, a0, sample)
    dt = time.perf_counter() - t0
    return {"value": tot / dt, "unit": "bins/s", "cores": 1, "kind": kind,
            "sample": "%d bins in %d trunks of <=500k, n=64, single thread, %.1f s" % (tot, len(sample), dt)}


def make_line(args, fast, world, bins, total_bins, lens, n_local_segs, kern, diag, value, ms_per_step, traj_src, n_moving, stats_numel):
    """The JSON line of the contract from what was measured (pure function: tests/test_bench_line.py feeds it fake
    measurements for every plan the library can report)."""
    # dominant KERNEL (one launch): the speculative forward / backward sweep or the expect kernel
    if fast:
        if diag.get("back_half") == 1:  # forward sweep, then backward sweep + counts in one kernel (estep_fused.hip)
            cand = {"k_fwd_struct<speculate>": kern["fwd_sweep"], "k_bwd_count4f_struct": kern["expect"]}
        elif diag.get("structured"):  # both bulk sweeps are ONE launch (k_sweep_struct): 2 x (8n+9) bytes per bin
            cand = {"k_sweep_struct": kern["fwd_sweep"], "k_expect_mfma": kern["expect"]}
        else:
            cand = {"k_fwd_fast<speculate>": kern["fwd_sweep"], "k_bwd_fast<speculate>": kern["bwd_sweep"], "k_expect_mfma": kern["expect"]}
    else:
        cand = {"k_fwd_exact": kern["forward"], "k_bwd_exact": kern["backward"], "k_expect_exact": kern["expect"]}
    dom = max(cand, key=lambda k: cand[k])
    dom_ms = cand[dom]
    # algorithmic HBM bytes per bin of each phase (forward writes the table and
    # the scale, the fused backward+expect reads them back; obs once per sweep)
    # counts: read X and bt (+ scales, obs); k_sweep_struct: write X and bt (+ scales), read obs twice; one sweep: half
    def alg_bytes(k):
        return (16 * N_STATES + 17) if k == "k_expect_mfma" else ((16 * N_STATES + 18) if k == "k_sweep_struct" else (8 * N_STATES + 9))
    alg_b = alg_bytes(dom)
    ach = bins * alg_b / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    fused = diag.get("back_half") == 1
    bytes_per_bin = (2 * (8 * N_STATES + 9)) if fused else BYTES_PER_BIN  # fused: X written once, read once; bt never stored
    pipe = bins * bytes_per_bin / (kern["total"] * 1e-3) / 1e9 if kern["total"] > 0 else 0.0
    # FP64 work of the fused kernel per bin: the counts (2 n^2 flop on v_mfma_f64_16x16x4) plus the O(n) backward step
    flop_b = 2 * N_STATES * N_STATES + 24 * N_STATES
    fused_ms = kern.get("expect", 0.0)
    tfl = bins * flop_b / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else 0.0
    traffic = None
    try:  # HBM bytes per launch from the rocprofv3 PMC pass of this same command (profiles/, scripts/gpu_pmc.sh)
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if abs(pj["bins"] - bins) <= 64 and dom in pj["kernels"]:
            traffic = pj["kernels"][dom]["hbm_bytes_per_launch"]
    except Exception:
        pass
    out = {
        "metric": "genome bins/sec through forward-backward (n=64)",
        "value": value, "unit": "bins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: whole-genome .psmcfa-like batch, %d bins x %d states in %d segments "
                               "%s, -p %s, one E-step (EM iteration) per step, parameters of a different EM round every step"
                               % (int(lens.sum()), N_STATES, len(lens), "per GPU" if args.scaling == "weak" else "sharded over the GPUs", PATTERN),
                   "mode": args.mode, "bins_per_gpu": bins, "bins_total": total_bins, "n_states": N_STATES, "segments": n_local_segs,
                   "longest_segment": int(lens.max()),
                   "parameters": ("fixed (n64_curve)" if args.fixed_params else "cycle of %d EM rounds: %s" % (n_moving, traj_src)),
                   "sharding": ("segments/GPU + 1 RCCL all-reduce(%d f64)/step" % stats_numel) if world > 1 else "single GPU",
                   **({"tiles": diag.get("n_chunks"), "speculative_overlap_bins": diag.get("warmup"),
                       "structured_sweeps": diag.get("structured"), "tile_bins": diag.get("tile_len"),
                       "sweep_items": [diag.get("items_fwd"), diag.get("items_bwd")],
                       "repair_rounds": [diag.get("fwd_rounds"), diag.get("bwd_rounds")],
                       "repaired_tiles": [diag.get("fwd_tiles"), diag.get("bwd_tiles")],
                       "boundary_err": max(diag.get("warm_err_fwd", 0), diag.get("warm_err_bwd", 0))} if diag else {})},
        "roofline": {**({"bound": "mfma", "kernel": dom, "launches_per_step": diag.get("fused_launches", 1),
                         "achieved": tfl, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / F64_PEAK_TFLOPS, "alg_flop_per_bin": flop_b,
                         "hbm": {"alg_bytes_per_bin": 8 * N_STATES + 9, "achieved_GBs": bins * (8 * N_STATES + 9) / (dom_ms * 1e-3) / 1e9,
                                 "alg_bytes_per_launch": bins * (8 * N_STATES + 9) / max(1, diag.get("fused_launches", 1))},
                         "note": "kernel_ms = the launches of one E-step summed (two-phase plan: tile lists A and B, half of the tiles each); "
                                 "traffic = PMC bytes of the larger launch"}
                        if dom == "k_bwd_count4f_struct" else
                        {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "alg_bytes_per_bin": alg_b, "alg_bytes_per_launch": bins * alg_b,
                         "note": "the longest single launch of the step.  Its first 45 % (the speculative warm-up of every tile) stores nothing and is "
                                 "bound by FP64 issue, shared with the other kernels of phase 1; the table stores all fall into the rest, where they run "
                                 "at the measured HBM write rate of this box (device_probes.hbm_GBs.sweep_store)"}),
                     # the other heavy kernel of the step, so that both rooflines are on the line whichever is longer
                     **({"also": {"bound": "mfma", "kernel": "k_bwd_count4f_struct", "launches_per_step": diag.get("fused_launches", 1), "kernel_ms": fused_ms,
                                  "achieved": tfl, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F64_PEAK_TFLOPS, "alg_flop_per_bin": flop_b,
                                  "mfma_cycles_frac": (bins / 4 * 16 * 64) / (fused_ms * 1e-3 * 1024 * 2.16e9) if fused_ms > 0 else None,
                                  "note": "backward sweep + counts, both launches summed; mfma_cycles_frac = 16 x 64-cycle v_mfma_f64_16x16x4 per step of four "
                                          "tiles over SIMD time at the 2.16 GHz the box sustains (SQ counters: profiles/r02_sq_counters.json)"}}
                        if fused and dom != "k_bwd_count4f_struct" else
                        ({"also": {"bound": "hbm", "kernel": "k_fwd_struct<speculate>", "kernel_ms": kern["fwd_sweep"], "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                   "achieved": bins * (8 * N_STATES + 9) / (kern["fwd_sweep"] * 1e-3) / 1e9,
                                   "frac": bins * (8 * N_STATES + 9) / (kern["fwd_sweep"] * 1e-3) / 1e9 / HBM_PEAK_GBS}} if fused and kern.get("fwd_sweep", 0) > 0 else {})),
                     "traffic": traffic, "kernel_ms": dom_ms,
                     "pipeline": {"alg_bytes_per_bin": bytes_per_bin, "ms": kern["total"], "achieved": pipe,
                                  "frac": pipe / HBM_PEAK_GBS},
                     "kernels_ms": kern,
                     "fp64_note": ("the backward sweep feeds the counts (K=bins GEMM, 2*n^2 flop/bin on v_mfma_f64_16x16x4) in the same "
                                   "wave; f64 matrix and vector instructions share the FP64 pipe (78.6 TFLOP/s dense either way): "
                                   "%.1f TFLOP/s of counts + O(n) sweep work in that kernel" if fused else
                                   "forward and backward sweep kernels run side by side and share the HBM; the counts "
                                   "kernel (K=bins GEMM, 2*n^2 flop/bin on v_mfma_f64_16x16x4) reaches %.1f of 78.6 TFLOP/s") %
                                  (bins * 2 * N_STATES * N_STATES / (kern["expect"] * 1e-3) / 1e12 if kern.get("expect", 0) > 0 else 0.0)},
    }
    return out


class Shard:
    """One rank's segments resident in HBM + the E-step context over them."""

    def __init__(self, hip, torch, segs, n_states, device, mode, opts):
        lens = np.array([len(s) for s in segs], dtype=np.int32)
        off = np.concatenate([[0], np.cumsum((lens.astype(np.int64) + 63) // 64 * 64)])
        host = np.full(int(off[-1]) + 256, 2, dtype=np.uint8)
        for s, o in zip(segs, off[:-1]):
            host[o:o + len(s)] = s
        self.d_obs = torch.from_numpy(host).cuda()
        self.es = hip.HipEStep(n_states, device=device, mode=mode)
        for kv in opts:
            k, v = kv.split("=")
            self.es.set_option(k, float(v))
        self.es.load_segments_device(self.d_obs.data_ptr(), off[:-1], lens, keepalive=self.d_obs)
        self.bins = int(lens.sum())
        self.stats = torch.zeros(n_states * n_states + 2 * n_states + 1, dtype=torch.float64, device="cuda")

    def close(self):
        self.es.close()
        self.es = None; self.d_obs = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--bins", type=int, default=30_000_000, help="bins per genome (whole human genome ~ 3e7)")
    ap.add_argument("--segments", type=int, default=90)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: one genome per GPU; strong: one genome, its segments LPT-sharded over the GPUs (config 3)")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--traj", default=os.path.join(ROOT, "tests", "golden", "traj_n64.json"))
    ap.add_argument("--fixed-params", type=int, default=0, help="1: the same parameters every step (steady state) as the headline")
    ap.add_argument("--cpu-sample", type=int, default=1_500_000, help="bins for the CPU baseline (0 = skip)")
    ap.add_argument("--exact-extra", type=int, default=1, help="also time 1 exact-mode step (0 = skip)")
    ap.add_argument("--n128-extra", type=int, default=1, help="also time config 5 (128 states, same genome) (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (chunk, warmup, ...)")
    args = ap.parse_args()

    import torch
    from psmc_amd import hip, sim
    from psmc_amd.dist import partition_segments

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # BENCH_SINGLE_GPU_TEST=1: exercise the N>1 code path on a 1-GPU box (all ranks on device 0, gloo)
    single_gpu_test = os.environ.get("BENCH_SINGLE_GPU_TEST") == "1"
    if single_gpu_test:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    a, e, a0 = load_params()
    traj, traj_src = load_trajectory(args.traj)
    lens = sim.human_like_lengths(args.bins, n_seg=args.segments)
    mode = hip.MODE_FAST if args.mode == "fast" else hip.MODE_EXACT
    stream = torch.cuda.current_stream()

    def workload(scaling):
        """(this rank's segments, bins of the whole job)"""
        t0 = time.perf_counter()
        if scaling == "weak" or world == 1:
            segs = sim.simulate_genome(a, e, a0, lens, seed=43 + rank)
            total = int(lens.sum()) * world
        else:  # one genome for the whole job; every rank draws it (1.8 s) and keeps its LPT share
            full = sim.simulate_genome(a, e, a0, lens, seed=43)
            mine = partition_segments(lens, world)[rank]
            segs = [full[i] for i in mine]
            total = int(lens.sum())
        log("[rank %d] %s: %d segments, %d bins here, longest %d (%.1f s)"
            % (rank, scaling, len(segs), sum(len(s) for s in segs), max(len(s) for s in segs), time.perf_counter() - t0))
        return segs, total

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def make_step(sh):
        def step(par):
            pa, pe, p0 = par
            if mode == hip.MODE_FAST:
                sh.es.estep_device(pa, pe, p0, sh.stats.data_ptr(), stream.cuda_stream)
                if dist is not None and not single_gpu_test:
                    dist.all_reduce(sh.stats)        # RCCL over xGMI: replaces hmm_add_expect across shards
                elif dist is not None:
                    t = sh.stats.cpu(); dist.all_reduce(t); sh.stats.copy_(t)
            else:
                r = sh.es.estep(pa, pe, p0)          # exact: ordered host sum (bit-identical to khmm.c)
                if dist is not None:
                    t = torch.from_numpy(np.concatenate([r["A"].ravel(), r["E"].ravel(), [r["LL"]]]))
                    t = t if single_gpu_test else t.cuda()
                    dist.all_reduce(t)
        return step

    def timed(step, params, steps, warmup):
        """K steps cycling through `params`, barrier + synchronize on both sides, MAX over ranks -> seconds."""
        for i in range(warmup):
            step(params[i % len(params)])
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(params[(warmup + i) % len(params)])
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if single_gpu_test else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    moving = traj if not args.fixed_params else [(a, e, a0)]
    segs, total_bins = workload(args.scaling)
    sh = Shard(hip, torch, segs, N_STATES, local, mode, args.opt)
    es = sh.es
    step = make_step(sh)
    first_ms = None
    if mode == hip.MODE_FAST:
        t0 = time.perf_counter()
        es.estep(*moving[0])   # blocking form once: allocates the tables, plans, learns the slow regions
        first_ms = (time.perf_counter() - t0) * 1e3
    dt = timed(step, moving, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = total_bins / (dt / args.steps)
    # per-kernel durations (HIP events recorded by the library on the streams the kernels ran on); measured on separate
    # steps of the same cycle so that the event reads do not perturb the timed region
    kern = {}
    nk = max(1, min(args.steps, len(moving))) if mode == hip.MODE_FAST else 1
    for i in range(nk):
        step(moving[i % len(moving)]); torch.cuda.synchronize()
        t = es.timing()
        for k in t:
            kern[k] = kern.get(k, 0.0) + t[k] / nk
    diag = es.fast_diag() if mode == hip.MODE_FAST else {}
    # steady state: the same parameters every step (the plan has seen them: no repairs, nothing to learn)
    steady = None
    if mode == hip.MODE_FAST and not args.fixed_params:
        dts = timed(step, [(a, e, a0)], args.steps, max(2, args.warmup // 2))
        steady = {"ms_per_step": dts / args.steps * 1e3, "value": total_bins / (dts / args.steps), "unit": "bins/s",
                  "note": "same (a, e, a0) every step -- what BENCH_r01 measured; the headline cycles through %d parameter sets" % len(moving)}
    bins = sh.bins

    out = None
    if rank == 0:
        out = make_line(args, mode == hip.MODE_FAST, world, bins, total_bins, lens, len(segs), kern, diag, value, ms_per_step, traj_src,
                        len(moving), sh.stats.numel())
        if steady is not None:
            out["steady_state"] = steady
        if first_ms is not None:
            out["first_call_ms"] = first_ms
        if world == 1:
            try:  # what plain kernels reach on this box (context for the fractions above; diagnostics of the library)
                hb = hip.hbm_probe(4 << 30, device=local)
                lp = hip.load_probe(2048, 8000, device=local)
                out["roofline"]["device_probes"] = {
                    "hbm_GBs": {k: round(v) for k, v in hb.items()},
                    "structured_step": {"waves": 2048, "cycles_per_step": round(lp["cycles_per_step"], 1), "shader_MHz_under_load": round(lp["mhz"])},
                    "note": "streaming fill/read/copy and the sweeps' store pattern; the O(N) step on 2 waves per SIMD (FP64 issue) and the clock it sustains"}
            except Exception as ex_:
                out["roofline"]["device_probes"] = {"error": str(ex_)}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(a, e, a0, segs, args.cpu_sample)
        if world == 1 and mode == hip.MODE_FAST and diag.get("structured"):
            try:  # the same E-step without the N x N counts: what the psmc binary uses with the O(N) objective
                for i in range(len(moving)):
                    es.estep_factored(*moving[i])
                t1 = time.perf_counter()
                nf = 2 * len(moving)
                for i in range(nf):
                    es.estep_factored(*moving[i % len(moving)])
                dtf = (time.perf_counter() - t1) / nf
                out["factored_stats"] = {"value": bins / dtf, "unit": "bins/s", "ms_per_step": dtf * 1e3, "kernels_ms": es.timing(),
                                         "note": "psmc_hip_estep_factored: triangular sums of A, E, LL from the backward sweep in O(N) "
                                                 "per bin (no counts GEMM, no bt table); blocking call incl. read-back; moving parameters"}
            except Exception as ex_:
                out["factored_stats"] = {"error": str(ex_)}
    # ---- config 3 proper beside a weak-scaling headline: ONE genome sharded over the ranks
    if world > 1 and args.scaling == "weak" and mode == hip.MODE_FAST:
        sh.close()
        segs_s, total_s = workload("strong")
        sh = Shard(hip, torch, segs_s, N_STATES, local, mode, args.opt)
        step = make_step(sh)
        sh.es.estep(*moving[0])
        dts = timed(step, moving, args.steps, args.warmup)
        if rank == 0:
            out["strong_scaling"] = {"value": total_s / (dts / args.steps), "unit": "bins/s", "ms_per_step": dts / args.steps * 1e3,
                                     "bins_total": total_s, "bins_this_rank": sh.bins,
                                     "note": "config 3: one %d-bin genome, segments LPT-sharded over %d GPUs, 1 all-reduce per step; "
                                             "MAX over ranks like the headline" % (total_s, world)}
    if rank == 0 and world == 1 and args.exact_extra > 0 and mode == hip.MODE_FAST:
        try:
            lens_l = np.array([len(s) for s in segs], dtype=np.int32)
            off = np.concatenate([[0], np.cumsum((lens_l.astype(np.int64) + 63) // 64 * 64)])
            d_obs = sh.d_obs
            sh.es.close()
            ex = hip.HipEStep(N_STATES, device=local, mode=hip.MODE_EXACT)
            ex.load_segments_device(d_obs.data_ptr(), off[:-1], lens_l, keepalive=d_obs)
            ex.estep(a, e, a0)
            t1 = time.perf_counter(); ex.estep(*moving[1 % len(moving)]); dte = time.perf_counter() - t1
            out["exact_mode"] = {"value": bins / dte, "unit": "bins/s", "ms_per_step": dte * 1e3,
                                 "kernels_ms": ex.timing(),
                                 "note": "bit-identical to khmm.c; one wave per segment, critical path = longest segment (%d bins)" % int(lens_l.max())}
            ex.close()
        except Exception as ex_:  # the headline number must survive an extra's failure
            out["exact_mode"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and args.n128_extra > 0 and mode == hip.MODE_FAST:
        try:  # config 5: -p "64*2", 128 states, the same genome
            g = np.load(os.path.join(ROOT, "tests", "golden", "estep_n128.npz"))
            a8, e8, a08 = g["n128_curve.a"], g["n128_curve.e"], g["n128_curve.a0"]
            sh.close()
            s8 = Shard(hip, torch, segs, 128, local, hip.MODE_FAST, args.opt)
            t1 = time.perf_counter(); s8.es.estep(a8, e8, a08); f8 = (time.perf_counter() - t1) * 1e3
            st8 = torch.zeros(128 * 128 + 2 * 128 + 1, dtype=torch.float64, device="cuda")
            for _ in range(4):  # the tile plan settles within three E-steps of the same parameters
                s8.es.estep_device(a8, e8, a08, st8.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                s8.es.estep_device(a8, e8, a08, st8.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            d8 = (time.perf_counter() - t1) / 5
            kern8 = s8.es.timing()
            fac8 = None
            try:  # the E-step without the 128 x 128 counts (psmc_hip_estep_factored: what `psmc -p 64*2` runs in fast mode)
                for _ in range(2):
                    s8.es.estep_factored(a8, e8, a08)
                t1 = time.perf_counter()
                for _ in range(5):
                    s8.es.estep_factored(a8, e8, a08)
                fac8 = {"ms_per_step": (time.perf_counter() - t1) / 5 * 1e3, "kernels_ms": s8.es.timing()}
                fac8["value"] = bins / (fac8["ms_per_step"] * 1e-3)
            except Exception as ex_:
                fac8 = {"error": str(ex_)}
            out["n128"] = {"value": bins / d8, "unit": "bins/s", "ms_per_step": d8 * 1e3, "first_call_ms": f8, "kernels_ms": kern8, "factored_stats": fac8,
                           "config": "configs[4]: -p 64*2 (128 states), %d bins in %d segments, fast mode, fixed parameters" % (bins, len(segs)),
                           "alg_bytes_per_bin": 16 * 128 + 18, "alg_flop_per_bin_counts": 2 * 128 * 128}
            s8.close()
        except Exception as ex_:
            out["n128"] = {"error": str(ex_)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
