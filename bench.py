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
    python bench.py --engine group --gpus N        # ONE process: the C library's own sharding (psmc_hip_group_*, what
                                                   # the psmc binary runs with PSMC_HIP_DEVICES=0,1,..): LPT in C, one host
                                                   # thread per device, RCCL all-reduce via group.hip

Extras on the same line (N=1): `shard_sweep` (the E-step on rank 0's LPT share of the genome at 1/2/4/8 GPUs and on the
500 k-bin input of config 2, with the strong-scaling curve those times predict), `group_engine` (the product's own
multi-GPU path, see above; with N>1 it is measured beside the torch.distributed number), `boot` (config 4 through the
psmc_boot binary, 16 replicates), `factored_stats`, `exact_mode`, `n128` (config 5).

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
ALLREDUCE_MS = 0.05                  # all-reduce of n*n+2n+1 doubles (34 KB) over xGMI: latency-bound; used only for the PREDICTED strong-scaling curve


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
    model = "unknown"
    try:   # the host CPU model and its core count beside the single-thread number
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    out = {"value": tot / dt, "unit": "bins/s", "cores": 1, "kind": kind, "host_cpu": model, "host_cores": os.cpu_count(),
           "sample": "%d bins in %d trunks of <=500k, n=64, single thread, %.1f s" % (tot, len(sample), dt)}
    if kind == "reference":   # which build of the reference this is: the GPU box cannot rebuild it
        import hashlib
        try:
            out["artefact"] = {"file": "oracle/_ref/libpsmc_ref.so", "sha256_16": hashlib.sha256(open(orc.REF_SO, "rb").read()).hexdigest()[:16],
                               "built_by": "`make -C oracle ref`: gcc -g -Wall -O2 (the reference's own flags) on its unmodified khmm.c kmin.c cli.c core.c em.c aux.c + oracle/ref_shim.c, in the build container; git-ignored, travels with the snapshot"}
        except OSError:
            pass
    try:
        out["multi"] = cpu_baseline_multi(a, e, a0, segs, kind, tot / dt)
    except Exception as ex_:
        out["multi"] = {"error": str(ex_)[-300:]}
    return out


def usable_cores():
    """what the box really gives this process: the scheduler affinity, capped by the cgroup's CPU quota (the driver's container: 16)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline_multi(a, e, a0, segs, kind, one_core_rate):
    """SURVEY section 8(d) "optionally a P-process CPU run": segments are independent (em.c:36-55), so the reference's honest
    ceiling on this host is P processes, one 500 k-bin trunk each, P = the cores this container may use -- and, whatever P,
    one E-step cannot finish before the longest segment has been swept by one core (VERDICT r4 missing 5)."""
    import subprocess as sp
    import tempfile
    P = usable_cores()
    trunks = []
    for s in segs:
        for j in range(0, len(s) - 499_999, 500_000):
            if len(trunks) < P:
                trunks.append(s[j:j + 500_000])
    P = len(trunks)
    d = tempfile.mkdtemp(prefix="psmc_cpu_multi_")
    np.savez(os.path.join(d, "in.npz"), a=a, e=e, a0=a0, **{"t%d" % i: t for i, t in enumerate(trunks)})
    code = ("import sys, time, numpy as np; sys.path.insert(0, %r); import orc; g = np.load(%r); "
            "eng = orc.Reference() if %r == 'reference' else orc.Oracle(); t0 = time.perf_counter(); "
            "eng.estep(g['a'], g['e'], g['a0'], [g['t' + sys.argv[1]]]); print(time.perf_counter() - t0)"
            % (os.path.join(ROOT, "tests"), os.path.join(d, "in.npz"), kind))
    t0 = time.perf_counter()
    procs = [sp.Popen([sys.executable, "-c", code, str(i)], stdout=sp.PIPE, stderr=sp.DEVNULL, text=True) for i in range(P)]
    times = [float(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    wall = time.perf_counter() - t0
    bins = sum(len(t) for t in trunks)
    longest = max(len(s) for s in segs)
    total = sum(len(s) for s in segs)
    rate = bins / max(times)
    return {"value": rate, "unit": "bins/s", "cores": P, "kind": kind,
            "sample": "%d processes x one 500 k-bin trunk at once: slowest %.2f s (wall incl. start-up %.2f s)" % (P, max(times), wall),
            "per_core_under_load": bins / sum(times),
            "estep_floor_s": {"longest_segment_bins": int(longest), "one_core": longest / one_core_rate,
                              "throughput_at_%d_cores" % P: total / rate,
                              "note": "one E-step of this workload on this host's CPU cannot take less than max(these two): segments are the "
                                      "only parallelism the reference's algorithm has (em.c:36-55)"}}


# the kernel sources a replayed counter record (profiles/pmc_traffic*.json, sq_factored.json) must have been taken from to be quoted
TRAFFIC_SOURCES = ["psmc_amd/csrc/estep_fused.hip", "psmc_amd/csrc/estep_struct.hip", "psmc_amd/csrc/estep_fast.hip", "psmc_amd/csrc/struct_prims.h", "psmc_amd/csrc/wave_prims.h"]


def kernel_src_sha16(sources, root=ROOT):
    import hashlib
    return hashlib.sha256(b"".join(open(os.path.join(root, s), "rb").read() for s in sources)).hexdigest()[:16]


def replayed_traffic(path, bins, kernel):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc record -- only when the record was taken from the kernel sources
    of THIS build (after a kernel change the line must not quote the old kernel's traffic).  -> (bytes or None, note)."""
    try:
        pj = json.load(open(path))
    except Exception:
        return None, "no PMC record (%s)" % os.path.basename(path)
    want = pj.get("kernel_src_sha16")
    if not want:
        return None, "%s carries no kernel_src_sha16: taken before round 6, cannot be matched to this build (scripts/lease.sh prof re-takes it)" % os.path.basename(path)
    have = kernel_src_sha16(pj.get("kernel_sources", TRAFFIC_SOURCES))
    if have != want:
        return None, "%s was taken from other kernel sources (sha %s, now %s): no traffic figure until it is re-taken (scripts/lease.sh prof)" % (os.path.basename(path), want, have)
    if abs(pj.get("bins", -1) - bins) > 64 or kernel not in pj.get("kernels", {}):
        return None, "%s has no entry for %s at %d bins" % (os.path.basename(path), kernel, bins)
    return pj["kernels"][kernel]["hbm_bytes_per_launch"], "replayed from %s (a separate rocprofv3 --pmc pass of this command, same kernel sources: sha %s)" % (os.path.basename(path), want)


def factored_roofline(bins, kern, dt_step):
    """the path `psmc` runs in fast mode has no matrix instruction and little memory traffic (130 B per bin with
    checkpoints); its kernels are bound by vector-instruction ISSUE.  One SIMD issues at most one f64 vector instruction per 4
    cycles, so the roof is SIMDs x sustained clock / 4 wave-instructions per second; `achieved` = the vector instructions the back
    half's kernel issues per launch (SQ_INSTS_VALU of a separate rocprofv3 --pmc pass, scaled by the counters' coverage of the
    device: profiles/sq_factored.json) over its launch duration measured live.  HBM beside it."""
    simds, clock = 1024, 2.1e9    # 256 CUs x 4; the shader clock these kernels sustain (scripts/sweep_trace.py: 1.8-2.2 GHz under load)
    peak = simds * clock / 4.0
    r = {"bound": "valu_issue", "kernel": "k_bwd_acc_ckpt", "kernel_ms": kern.get("expect"), "peak": peak, "unit": "wave-instructions/s",
         "hbm": {"alg_bytes_per_bin": 2 * (8 * N_STATES + 9) / 8.0, "achieved_GBs": bins * 2 * (8 * N_STATES + 9) / 8.0 / dt_step / 1e9,
                 "frac": bins * 2 * (8 * N_STATES + 9) / 8.0 / dt_step / 1e9 / HBM_PEAK_GBS},
         "achieved": None, "frac": None, "valu_per_bin": None}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "sq_factored.json")))
        k = pj["kernels"]["k_bwd_acc_ckpt"]
        # the instruction counts are a STATIC record: valid only for the kernel sources they were taken from
        import hashlib
        sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, s), "rb").read() for s in pj.get("kernel_sources", []))).hexdigest()[:16]
        r["counts_valid"] = bool(pj.get("kernel_src_sha16")) and sha == pj.get("kernel_src_sha16")
        r["clock_assumed_GHz"] = clock / 1e9
        if not r["counts_valid"]:
            r["note"] = "profiles/sq_factored.json was taken from other kernel sources (sha %s, now %s): no issue roofline until it is re-taken (scripts/lease.sh prof)" % (pj.get("kernel_src_sha16"), sha)
        elif abs(pj["bins"] - bins) <= 64 and kern.get("expect", 0) > 0:
            r["valu_per_bin"] = k["valu_per_launch"] / pj["bins"]
            r["achieved"] = k["valu_per_launch"] / (kern["expect"] * 1e-3)
            r["frac"] = r["achieved"] / peak
            r["forward_sweep"] = {"kernel": "k_fwd_struct<ckpt>", "kernel_ms": kern.get("fwd_sweep"),
                                  "valu_per_bin": pj["kernels"]["k_fwd_struct<false,4,true>"]["valu_per_launch"] / pj["bins"]}
            r["note"] = pj.get("note", "")
    except Exception:
        pass
    return r


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
    # HBM bytes per launch from the rocprofv3 PMC pass of this same command (profiles/, scripts/lease.sh prof), guarded by the sources' hash
    traffic, traffic_note = replayed_traffic(os.path.join(ROOT, "profiles", "pmc_traffic.json"), bins, dom)
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
                                 "traffic = PMC bytes of the larger launch, replayed from profiles/pmc_traffic.json (a separate rocprofv3 --pmc pass of this "
                                 "command: PMC counters cannot be read inside the timed run)"}
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
                     "traffic": traffic, "traffic_note": traffic_note, "kernel_ms": dom_ms,
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


def median_ms(fn, n, sync):
    """median / min wall time in ms of n calls of fn(i), each followed by sync()"""
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); fn(i); sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(min(ts))


def shard_sweep_extra(hip, torch, sim, partition_segments, full, lens, a, e, a0, moving, device, opts, stream, t_full_ms):
    """the E-step on shard-sized inputs.  Rank 0's LPT share of the benchmark genome at N = 2, 4, 8 (the
    largest share: rank 0 holds the longest segment) and the 500 k-bin single segment of config 2, parameters moving every
    step like the headline; predicted strong scaling = t(1) / (t(share_N) + all-reduce latency)."""
    res = {"note": "fast mode, full counts, moving parameters, median of 12 steps after 8; predicted_speedup = headline ms / (share ms + %.2f ms all-reduce of 34 KB, "
                   "latency-bound on xGMI); what the driver's N>1 strong-scaling run should show if the shards are balanced" % ALLREDUCE_MS, "workloads": []}
    work = []
    rng = np.random.default_rng(7)
    work.append(("config2_chr22like_500k", 1, [sim.simulate_segment(a, e, a0, 500_000, rng)]))
    for n in (8, 4, 2):
        mine = partition_segments(lens, n)[0]
        work.append(("genome_share_1of%d" % n, n, [full[i] for i in mine]))
    for name, n, segs in work:
        sh = Shard(hip, torch, segs, N_STATES, device, hip.MODE_FAST, opts)
        try:
            sh.es.estep(*moving[0])
            run = lambda i: sh.es.estep_device(*moving[i % len(moving)], sh.stats.data_ptr(), stream.cuda_stream)
            for i in range(8):
                run(i)
            torch.cuda.synchronize()
            med, mn = median_ms(lambda i: run(8 + i), 12, torch.cuda.synchronize)
            d = sh.es.fast_diag()
            r = {"workload": name, "bins": sh.bins, "segments": len(segs), "ms_per_step": med, "ms_min": mn, "bins_per_s": sh.bins / (med * 1e-3),
                 "tiles": d["n_chunks"], "tile_bins": d["tile_len"], "merged_phase1": d["merged_phase1"], "fused_launches": d["fused_launches"],
                 "kernels_ms": {k: round(float(v), 3) for k, v in sh.es.timing().items()}}
            if name.startswith("genome_share"):
                r["n_gpus"] = n
                r["predicted_speedup"] = t_full_ms / (med + ALLREDUCE_MS)
                r["predicted_efficiency"] = r["predicted_speedup"] / n
        except Exception as ex_:
            r = {"workload": name, "error": str(ex_)}
        res["workloads"].append(r)
        sh.close()
    return res


def real_shape_extra(hip, torch, full, moving, device, opts, stream, t_full_ms):
    """every timed input is drawn from the model with missing runs of 10-90 bins; a real .psmcfa carries centromere and
    assembly-gap runs of 1e4 .. 3e5 `N` bins (utils/fq2psmcfa.c:114-127).  The same genome with such gaps planted -- one run of 30,000 bins in
    each of the 22 chromosomes, 180,000 in the longest (chr1's heterochromatin), 3,000 at both ends of every chromosome: 3.4 % of the bins --
    timed like the headline.  Until round 5 a gap longer than "group_cap" cost dozens of repair rounds in EVERY E-step ("gap_tiles")."""
    segs = [s.copy() for s in full]
    planted = 0
    for i, s in enumerate(segs[:22]):
        L = len(s)
        if L < 200_000:
            continue
        c0 = int(L * 0.42)
        n = 180_000 if i == 0 else 30_000
        s[c0:c0 + n] = 2; s[:3000] = 2; s[-3000:] = 2
        planted += n + 6000
    sh = Shard(hip, torch, segs, N_STATES, device, hip.MODE_FAST, opts)
    try:
        sh.es.estep(*moving[0])
        run = lambda i: sh.es.estep_device(*moving[i % len(moving)], sh.stats.data_ptr(), stream.cuda_stream)
        for i in range(10):
            run(i)
        torch.cuda.synchronize()
        med, mn = median_ms(lambda i: run(10 + i), 12, torch.cuda.synchronize)
        d = sh.es.fast_diag(); pl = sh.es.fast_plan()
        r = {"workload": "the benchmark genome with %d bins of planted gaps (22 x 30 k-bin centromeres, 180 k in the longest segment, 3 k-bin telomeres)" % planted,
             "bins": sh.bins, "ms_per_step": med, "ms_min": mn, "vs_headline": med / t_full_ms, "repair_rounds_last_step": [d["fwd_rounds"], d["bwd_rounds"]],
             "tiles": pl["tiles"], "glued": [pl["glued_fwd"], pl["glued_bwd"]], "kernels_ms": {k: round(float(v), 3) for k, v in sh.es.timing().items()}}
    except Exception as ex_:
        r = {"error": str(ex_)}
    sh.close()
    return r


def group_engine_run(hip, segs, devices, moving, steps, warmup, mode):
    """The product's own multi-GPU path (psmc_hip_group_*, group.hip): one process, LPT partition in C, one host thread per
    device, RCCL all-reduce (or the host sum when devices repeat) -- selfcheck first, so that a failure names itself."""
    g = hip.HipGroup(N_STATES, devices, mode=mode)
    try:
        sc = g.selfcheck()
        g.load_segments(segs)
        t0 = time.perf_counter(); g.estep(*moving[0]); first = (time.perf_counter() - t0) * 1e3
        for i in range(warmup):
            g.estep(*moving[i % len(moving)])
        t0 = time.perf_counter()
        for i in range(steps):
            g.estep(*moving[(warmup + i) % len(moving)])
        dt = time.perf_counter() - t0
        info = g.info()
        return {"selfcheck": sc, "ms_per_step": dt / steps * 1e3, "first_call_ms": first, "steps": steps, "warmup": warmup, "devices": list(devices),
                "reduce": {0: "single shard", 1: "RCCL all-reduce (group.hip)", 2: "host sum in shard order", 3: "ordered per-segment sum"}[info["last_reduce"]]}
    finally:
        g.close()


def boot_extra(n_rep=16, iters=3):
    """Config 4 through the product binary: psmc_boot -R 16 -- -N3 over the splitfa trunks of the benchmark genome
    (utils/splitfa.c:20-35: 500 k-bin trunks; scripts/northstar_data.py), fast and exact mode, per-iteration times from
    PSMC_TIMING; and the same exact job with the main run beside it (psmc_boot --main: README:49-62 as one job).
    The fast run comes FIRST: the device memory a process leaves behind is cleared by the driver when the next process
    allocates it -- after an exact batch (250 GB of tables) that costs the next process ~4 s inside its first large
    hipMalloc, which round 4's record showed as a 5.7 s first iteration of the fast run (profiles/r05_boot_first_iteration.txt)."""
    import re, subprocess, tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import northstar_data as nd
    host = os.path.join(ROOT, "psmc_amd", "host")
    tmp = tempfile.mkdtemp(prefix="psmc_bench_")
    fd = nd.files(tmp)
    res = {"workload": "%d trunks, %d bins, longest %d; %d replicates, -N%d -t15 -r5 -p %s" % (fd["n_trunks"], fd["trunk_bins"], fd["longest_trunk"], n_rep, iters, PATTERN),
           "note": "psmc_boot binary (psmc_hip_estep_batch_cb under it): the E-steps of a context's replicates as one batch on the device, each replicate's M-step on a host thread as soon as the batch reports it final; msteps = what is left of them after the last batch; "
                   "per_iteration_ms covers ALL replicates; exact_with_main: the same plus the main run of the 90-segment genome on a thread of its own (--main)"}
    try:
        for m in ("fast", "exact", "exact_with_main"):
            env = dict(os.environ, PSMC_HIP_MODE=m.split("_")[0], PSMC_TIMING="1")
            cmd = [os.path.join(host, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, "b_%s-%%d.psmc" % m)]
            if m == "exact_with_main":
                cmd += ["--main", os.path.join(tmp, "b_main.psmc"), "--main-input", fd["genome"]]
            cmd += ["--", "-N%d" % iters, "-t15", "-r5", "-p", PATTERN, fd["split"]]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
            wall = time.perf_counter() - t0
            its = [(float(x.group(1)), float(x.group(2))) for x in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms", r.stderr)]
            res[m] = {"rc": r.returncode, "wall_s": round(wall, 2), "per_iteration_ms": [{"esteps": x, "msteps": y} for x, y in its],
                      "esteps_ms_per_replicate_last": (its[-1][0] / n_rep) if its else None}
            mes = [float(x.group(1)) for x in re.finditer(r"\[psmc\] E-step ([0-9.]+) ms", r.stderr)]
            if mes:
                res[m]["main_run_estep_ms"] = mes
            if r.returncode != 0:
                res[m]["stderr_tail"] = r.stderr[-300:]
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return res


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
    ap.add_argument("--engine", default="dist", choices=["dist", "group"],
                    help="dist: one process per GPU, torch.distributed (RCCL) all-reduce; group: ONE process, psmc_hip_group_* over --gpus devices (the psmc binary's path)")
    ap.add_argument("--group-devices", default="", help="engine group: explicit device list, e.g. 0,0 (two shards on one GPU, for testing)")
    ap.add_argument("--shard-extra", type=int, default=1, help="also time shard-sized inputs + the predicted strong-scaling curve (0 = skip)")
    ap.add_argument("--group-extra", type=int, default=1, help="also time the C library's own multi-GPU engine beside the headline (0 = skip)")
    ap.add_argument("--boot-extra", type=int, default=1, help="also run config 4 (16 bootstrap replicates, -N2) through psmc_boot (0 = skip)")
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
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        # First contact: does the exchange the timed steps will use see every rank, and which devices are they on?  (VERDICT r5 item 3c:
        # a SCALE record must answer "did RCCL see N ranks" by itself.)  One all-reduce of ones, one gather of the ranks' devices.
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if single_gpu_test else "cuda")
        dist.all_reduce(ones)
        props = torch.cuda.get_device_properties(local)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "device": local, "name": props.name, "pci_bus": getattr(props, "pci_bus_id", None)})
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_in_allreduce": int(round(float(ones.item()))),
                "rank_devices": seen, "distinct_devices": len({(d["device"], d["pci_bus"]) for d in seen})}
        try:
            comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if comm["backend"] == "nccl" else None
        except Exception:
            comm["rccl_version"] = None

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
    if args.engine == "group":
        # ONE process drives all devices through the C library (what `PSMC_HIP_DEVICES=0,1,.. psmc` does): no torch.distributed
        if world != 1:
            raise SystemExit("--engine group is one process: start it with plain `python bench.py --engine group --gpus N`")
        devices = [int(x) for x in args.group_devices.split(",")] if args.group_devices else list(range(args.gpus))
        if args.scaling == "weak":   # one genome per device, dealt by the library's LPT like any other segment list
            segs = [s for r in range(len(devices)) for s in sim.simulate_genome(a, e, a0, lens, seed=43 + r)]
        else:
            segs = sim.simulate_genome(a, e, a0, lens, seed=43)
        total_bins = int(sum(len(x) for x in segs))
        res = group_engine_run(hip, segs, devices, moving, args.steps, args.warmup, mode)
        ms = res["ms_per_step"]
        gbs = total_bins * BYTES_PER_BIN / (ms * 1e-3) / 1e9
        out = {"metric": "genome bins/sec through forward-backward (n=64)", "value": total_bins / (ms * 1e-3), "unit": "bins/s", "n_gpus": len(devices),
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[2]: whole-genome .psmcfa-like batch, %d bins x %d states in %d segments %s, -p %s, one E-step per step, "
                                      "parameters of a different EM round every step" % (int(lens.sum()), N_STATES, len(lens),
                                                                                       "per GPU" if args.scaling == "weak" else "sharded over the GPUs", PATTERN),
                          "engine": "group: one process, psmc_hip_group_* (psmc_amd/csrc/group.hip): LPT partition in C, one host thread per device, blocking call incl. the "
                                    "34 KB read-back", "mode": args.mode, "bins_total": total_bins, "n_states": N_STATES, "segments": len(segs), "devices": devices,
                          "sharding": res["reduce"], "selfcheck": res["selfcheck"]},
               "roofline": {"bound": "hbm", "kernel": "whole E-step (per-kernel rooflines: --engine dist)", "achieved": gbs, "peak": HBM_PEAK_GBS * len(set(devices)), "unit": "GB/s",
                            "frac": gbs / (HBM_PEAK_GBS * len(set(devices))), "alg_bytes_per_bin": BYTES_PER_BIN, "traffic": None},
               "first_call_ms": res["first_call_ms"]}
        if len(devices) == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(a, e, a0, segs, args.cpu_sample)
        print(json.dumps(out), flush=True)
        return
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
        if comm is not None:
            out["config"]["comm"] = comm   # the exchange behind "sharding": backend, the ranks one all-reduce of ones counted, every rank's device
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
        # (the CPU baseline runs LAST, after every GPU extra: sixteen busy host processes right before a timed GPU extra cost it 10 %)
        if world == 1 and mode == hip.MODE_FAST and diag.get("structured"):
            try:  # the same E-step without the N x N counts: what the psmc binary uses with the O(N) objective
                for i in range(len(moving)):
                    es.estep_factored(*moving[i])
                t1 = time.perf_counter()
                nf = 2 * len(moving)
                for i in range(nf):
                    es.estep_factored(*moving[i % len(moving)])
                dtf = (time.perf_counter() - t1) / nf
                kf = es.timing()
                out["factored_stats"] = {"value": bins / dtf, "unit": "bins/s", "ms_per_step": dtf * 1e3, "kernels_ms": kf,
                                         "roofline": factored_roofline(bins, kf, dtf),
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
    if world > 1 and args.group_extra > 0 and mode == hip.MODE_FAST:
        # the C library's own sharding over the same N devices (what the psmc binary runs with PSMC_HIP_DEVICES=0..N-1; the
        # driver's scaling run otherwise only sees the Python path).  Rank 0 starts `bench.py --engine group` as a CHILD
        # process under a timeout: this path has never met more than one device, and a hang or a crash in it must cost this
        # extra, not the bench line.  The other ranks free their devices and wait on a key of the rendezvous store (host side) --
        # an RCCL barrier would leave a spinning kernel on the very devices the child measures.
        import datetime
        sh.close()
        sync()
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
                   and not k.startswith("TORCHELASTIC")}
            ge = {}
            for sc_name in ("weak", "strong"):   # (everything that can fail is inside the try: the other ranks wait for the key below)
                cmd = [sys.executable, os.path.abspath(__file__), "--engine", "group", "--gpus", str(world), "--scaling", sc_name, "--steps", str(args.steps),
                       "--warmup", str(args.warmup), "--cpu-sample", "0", "--bins", str(args.bins), "--segments", str(args.segments), "--traj", args.traj]
                if single_gpu_test:   # BENCH_SINGLE_GPU_TEST: the shards share device 0 (host sum instead of RCCL)
                    cmd += ["--group-devices", ",".join(["0"] * world)]
                try:
                    r_ = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
                    if r_.returncode != 0:
                        raise RuntimeError("rc %d: %s" % (r_.returncode, r_.stderr.strip()[-300:]))
                    j_ = json.loads(r_.stdout.strip().splitlines()[-1])
                    ge[sc_name] = {"value": j_["value"], "ms_per_step": j_["ms_per_step"], "first_call_ms": j_["first_call_ms"], "bins_total": j_["config"]["bins_total"],
                                   "devices": j_["config"]["devices"], "reduce": j_["config"]["sharding"], "selfcheck": j_["config"]["selfcheck"]}
                except Exception as ex_:
                    ge[sc_name] = {"error": str(ex_)[-400:]}
            ge["note"] = ("a child process of rank 0, `bench.py --engine group --gpus %d`: psmc_hip_group_* over devices 0..%d, RCCL all-reduce inside group.hip; "
                          "compare weak with the headline, strong with strong_scaling" % (world, world - 1))
            out["group_engine"] = ge
            store.set("psmc_bench_group_extra_done", "1")
        else:
            store.wait(["psmc_bench_group_extra_done"], datetime.timedelta(minutes=20))
    if rank == 0 and world == 1 and args.exact_extra > 0 and mode == hip.MODE_FAST:
        try:
            lens_l = np.array([len(s) for s in segs], dtype=np.int32)
            off = np.concatenate([[0], np.cumsum((lens_l.astype(np.int64) + 63) // 64 * 64)])
            d_obs = sh.d_obs
            pm = moving[1 % len(moving)]
            r_fast = sh.es.estep(*pm)            # what the timed path computes, at the parameters the exact E-step below is run with
            f_fast = sh.es.estep_factored(*pm)
            sh.es.close()
            ex = hip.HipEStep(N_STATES, device=local, mode=hip.MODE_EXACT)
            ex.load_segments_device(d_obs.data_ptr(), off[:-1], lens_l, keepalive=d_obs)
            ex.estep(a, e, a0)
            t1 = time.perf_counter(); r_ex = ex.estep(*pm); dte = time.perf_counter() - t1
            from psmc_amd.parity import fast_error_metrics
            fm = fast_error_metrics(r_fast, r_ex, pm[0], pm[1])
            lo_, up_ = np.tril(r_ex["A"], -1), np.triu(r_ex["A"], 1)
            ts_ = np.stack([lo_.sum(1), up_.sum(1), np.diag(r_ex["A"]).copy(), lo_.sum(0), up_.sum(0)])
            fm["factored_sums_max"] = float(np.abs(f_fast["sums"] - ts_).max() / np.abs(ts_).max())
            big_ = ts_ >= 1e-6 * ts_.max()
            fm["factored_sums_cell"] = float((np.abs(f_fast["sums"] - ts_)[big_] / ts_[big_]).max())
            out["exact_mode"] = {"value": bins / dte, "unit": "bins/s", "ms_per_step": dte * 1e3,
                                 "kernels_ms": ex.timing(),
                                 "note": "bit-identical to khmm.c; one wave per segment, critical path = longest segment (%d bins)" % int(lens_l.max()),
                                 "fast_vs_exact": dict(fm, note="the timed (fast) E-step against this exact one at the same parameters, full size: A_max / E_max = "
                                                                "max|x - ref| / max|ref| (the gate, 1e-10); A_cell / E_cell = largest relative error of a cell >= 1e-6 x the "
                                                                "largest; QA / QE = relative error of sum A log a and sum E log e, what hmm_Q reads (psmc_amd/parity.py)")}
            ex.close()
        except Exception as ex_:  # the headline number must survive an extra's failure
            out["exact_mode"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and args.n128_extra > 0 and mode == hip.MODE_FAST:
        try:  # config 5: -p "64*2", 128 states, the same genome
            g = np.load(os.path.join(ROOT, "tests", "golden", "estep_n128.npz"))
            fixed8 = (g["n128_curve.a"], g["n128_curve.e"], g["n128_curve.a0"])
            tj8 = os.path.join(ROOT, "tests", "golden", "traj_n128.json")
            if os.path.exists(tj8) and not args.fixed_params:   # parameters of consecutive EM rounds of `psmc -N25 -p 64*2` on this workload
                from psmc_amd import hostlib
                t8 = json.load(open(tj8))
                mov8 = [hostlib.hmm_params(t8["pattern"], r["params"]) for r in t8["rounds"] if r["round"] >= 1][:25]
                par8 = "cycle of %d EM rounds: %s" % (len(mov8), t8.get("source", "tests/golden/traj_n128.json"))
            else:
                mov8, par8 = [fixed8], "fixed (n128_curve)"
            sh.close()
            s8 = Shard(hip, torch, segs, 128, local, hip.MODE_FAST, args.opt)
            t1 = time.perf_counter(); s8.es.estep(*mov8[0]); f8 = (time.perf_counter() - t1) * 1e3
            st8 = torch.zeros(128 * 128 + 2 * 128 + 1, dtype=torch.float64, device="cuda")
            NW8, NT8 = 8, 9   # the 5-step mean after 4 warm-ups still held a learning round (35.8 ms in the driver's run against 27 in ours)
            run8 = lambda i: s8.es.estep_device(*mov8[i % len(mov8)], st8.data_ptr(), stream.cuda_stream)
            for i in range(NW8):
                run8(i)
            torch.cuda.synchronize()
            d8, d8min = median_ms(lambda i: run8(NW8 + i), NT8, torch.cuda.synchronize)
            kern8 = {}
            for i in range(4):
                run8(NW8 + NT8 + i); torch.cuda.synchronize()
                for k, v in s8.es.timing().items():
                    kern8[k] = kern8.get(k, 0.0) + v / 4
            diag8 = s8.es.fast_diag()
            fac8 = None
            try:  # the E-step without the 128 x 128 counts (psmc_hip_estep_factored: what `psmc -p 64*2` runs in fast mode)
                for i in range(NW8):
                    s8.es.estep_factored(*mov8[i % len(mov8)])
                fm, fmin = median_ms(lambda i: s8.es.estep_factored(*mov8[(NW8 + i) % len(mov8)]), NT8, lambda: None)
                fac8 = {"ms_per_step": fm, "ms_min": fmin, "kernels_ms": s8.es.timing(), "value": bins / (fm * 1e-3),
                        "note": "blocking call incl. read-back; median of %d after %d warm-up calls" % (NT8, NW8)}
            except Exception as ex_:
                fac8 = {"error": str(ex_)}
            # roofline of the back half on the ALGORITHMIC flops: 2 n^2 (counts, v_mfma_f64_16x16x4) + 24 n (one O(n) backward sweep) per bin.
            # k_bwd_count8x_struct (round 4: sixteen tiles per work-group, one sweep per tile, operands exchanged through LDS) executes
            # exactly that
            flop8 = 2 * 128 * 128 + 24 * 128
            kname8 = "k_bwd_count8x_struct"
            tf8 = bins * flop8 / (kern8["expect"] * 1e-3) / 1e12 if kern8.get("expect", 0) > 0 else 0.0
            traffic8, traffic8_note = replayed_traffic(os.path.join(ROOT, "profiles", "pmc_traffic_n128.json"), bins, kname8)
            out["n128"] = {"value": bins / (d8 * 1e-3), "unit": "bins/s", "ms_per_step": d8, "ms_min": d8min, "first_call_ms": f8, "kernels_ms": kern8, "factored_stats": fac8,
                           "config": "configs[4]: -p 64*2 (128 states), %d bins in %d segments, fast mode; parameters: %s; median of %d steps after %d warm-up steps"
                                     % (bins, len(segs), par8, NT8, NW8),
                           "tiles": diag8.get("n_chunks"), "tile_bins": diag8.get("tile_len"), "repair_rounds": [diag8.get("fwd_rounds"), diag8.get("bwd_rounds")],
                           "alg_bytes_per_bin": 16 * 128 + 18, "alg_flop_per_bin_counts": 2 * 128 * 128,
                           "roofline": {"bound": "mfma", "kernel": kname8, "launches_per_step": diag8.get("fused_launches", 1), "kernel_ms": kern8.get("expect"),
                                        "achieved": tf8, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf8 / F64_PEAK_TFLOPS, "alg_flop_per_bin": flop8,
                                        "mfma_only_frac": bins * 2 * 128 * 128 / (kern8["expect"] * 1e-3) / 1e12 / F64_PEAK_TFLOPS if kern8.get("expect", 0) > 0 else None,
                                        "hbm": {"alg_bytes_per_bin": 8 * 128 + 9, "achieved_GBs": bins * (8 * 128 + 9) / (kern8["expect"] * 1e-3) / 1e9 if kern8.get("expect", 0) > 0 else None},
                                        "traffic": traffic8, "traffic_note": traffic8_note,
                                        "executed_flop_per_bin": flop8,
                                        "note": "frac is on the algorithmic flops 2 n^2 + 24 n (VERDICT r3 item 2); mfma_only_frac counts the 2 n^2 alone; "
                                                "traffic = PMC bytes per launch from profiles/pmc_traffic_n128.json (a separate rocprofv3 --pmc pass of this command), null if absent"}}
            s8.close()
        except Exception as ex_:
            out["n128"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and mode == hip.MODE_FAST and args.shard_extra > 0 and not args.fixed_params:
        try:  # (after the extras that need the headline's context: every timing extra gets the device to itself)
            try:
                sh.close()
            except Exception:
                pass
            out["shard_sweep"] = shard_sweep_extra(hip, torch, sim, partition_segments, segs, lens, a, e, a0, moving, local, args.opt, stream, ms_per_step)
        except Exception as ex_:
            out["shard_sweep"] = {"error": str(ex_)}
        try:
            out["real_shape"] = real_shape_extra(hip, torch, segs, moving, local, args.opt, stream, ms_per_step)
        except Exception as ex_:
            out["real_shape"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and args.group_extra > 0 and mode == hip.MODE_FAST:
        try:  # the product's own multi-GPU engine on this one device: must reproduce the headline (within 2 %).
            # Every other context of the process is closed first: beside the headline's context and what the other extras left
            # behind, this engine measured 3-7 % slower than alone
            try:
                sh.close()
            except Exception:
                pass
            ge = group_engine_run(hip, segs, [local], moving, args.steps, max(args.warmup, 10), mode)   # a fresh context: past its learning steps
            ge["value"] = total_bins / (ge["ms_per_step"] * 1e-3); ge["vs_headline"] = ms_per_step / ge["ms_per_step"]
            out["group_engine"] = ge
        except Exception as ex_:
            out["group_engine"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and args.boot_extra > 0 and mode == hip.MODE_FAST:
        try:  # config 4 through the product binary (every context of this process is closed: the exact batch sizes its groups by the free memory)
            try:
                sh.close()
            except Exception:
                pass
            torch.cuda.empty_cache()
            out["boot"] = boot_extra()
        except Exception as ex_:
            out["boot"] = {"error": str(ex_)}
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(a, e, a0, segs, args.cpu_sample)
        except Exception as ex_:
            out["cpu_baseline"] = {"error": str(ex_)[-300:]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
