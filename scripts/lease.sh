#!/bin/bash
# One parametrised GPU-lease script (replaces the single-use gpu_r02_*.sh / r03/*.sh of the build sessions).  Run it on the GPU
# box:  gpurun --timeout N -- 'bash scripts/lease.sh <task> [args]'.  Everything is written under gpurun_out/.
#   suite [pytest args]           the GPU test suite (PSMC_HIP_POISON=vary in the environment: every device allocation poisoned)
#   bench [bench.py args]         the driver's bench command (default: --steps 20 --warmup 5) + a digest of the line
#   sweep CFG [CFG ...]           scripts/shard_sweep.py, one --cfg per argument ("" = defaults); env SHARES (8,4,2,1), CHR (500000),
#                                 REPEAT (1: A B A B ... when > 1), FACTORED (0), STATES (64; 128 = config 5), WARMUP (10), STEPS (12)
#   timeline SHARES CHR           rocprofv3 --kernel-trace of a few E-steps, kernel timeline of the last one (every kernel); env CFG (options), FACTORED, TAG (output suffix)
#   prof                          rocprofv3 kernel stats + PMC FETCH_SIZE / WRITE_SIZE passes (calibrated on a known copy) of the bench
#                                 command and of config 5 alone, SQ instruction counters of both (scripts/sq_summary.py);
#                                 then locally: python scripts/prof_summary.py <tag> 30000001 (and PROF_NAME=n128 ...), python scripts/sq_summary.py <tag> n64|n128 -> profiles/
#   wtrace [SHARE] ["opts"]       per-wave time stamps of phase 1: rebuilds the library with -DPSMC_TRACE_SWEEP, runs scripts/sweep_trace.py, rebuilds it plain
#   probes                        psmc_hip_pipe_probe2 table + psmc_hip_place_probe grid (scripts/probes.py)
#   trace [SHARE] ["opts"]        per-E-step trace over 30 moving-parameter E-steps: ms, repair rounds, plan (scripts/estep_trace.py)
#   northstar                     scripts/northstar.py: psmc_boot --main (main run + 100 bootstraps at -N25 as ONE job), exact and fast; NS_SEPARATE=1 (default
#                                 here): also `psmc` alone in both modes, and the joint job's main output byte for byte against it (~7 min)
#   final                         suite + bench + prof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/prof gpurun_out/pmc
export TMPDIR=/tmp
task=${1:-final}; shift || true

digest() { python - "$1" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.3f ms  %.3e bins/s  frac %.3f  steady %s" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r.get("steady_state", {}).get("ms_per_step")))
if "kernels_ms" in r["roofline"]: print("kernels", {k: round(v, 2) for k, v in r["roofline"]["kernels_ms"].items()})
print("factored", r.get("factored_stats", {}).get("ms_per_step"))
for w in r.get("shard_sweep", {}).get("workloads", []): print("shard", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in w.items() if k != "kernels_ms"})
print("real_shape", {k: v for k, v in r.get("real_shape", {}).items() if k not in ("workload", "kernels_ms")})
print("group", r.get("group_engine")); print("boot", json.dumps(r.get("boot"))[:1200])
n = r.get("n128", {})
print("n128", n.get("ms_per_step"), n.get("ms_min"), n.get("factored_stats", {}).get("ms_per_step") if isinstance(n.get("factored_stats"), dict) else n.get("factored_stats"), n.get("roofline", {}).get("frac"), n.get("error"))
print("exact", r.get("exact_mode", {}).get("ms_per_step"), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
}
show_sweep() { python - "$1" <<'PY'
import json, sys, collections
agg = collections.defaultdict(list)
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]; pl = r["plan"]; agg[(r["workload"], r["cfg"])].append(r["ms_median"])
    print("%-16s %-34s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f cnt %.2f  warm %4.0f/%4.0f glued %d/%d" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"],
          k["total"], k["fwd_sweep"], k["expect"], pl["warm_fwd_mean"], pl["warm_bwd_mean"], pl["glued_fwd"], pl["glued_bwd"]))
if any(len(v) > 1 for v in agg.values()):
    for k, v in sorted(agg.items(), key=lambda x: (x[0][0], sum(x[1]) / len(x[1]))): print("== %-16s %-34s mean %.3f  %s" % (k[0], k[1], sum(v) / len(v), [round(x, 2) for x in v]))
PY
}
do_suite() {
  timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/pytest_gpu.log | head -12 | cut -c1-250
}
do_bench() {
  [ $# -eq 0 ] && set -- --steps 20 --warmup 5
  s=$(date +%s); timeout 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; tail -2 gpurun_out/bench.err | cut -c1-300
  digest gpurun_out/bench.json
}
do_prof() {
  rm -rf $R/gpurun_out/prof/bench* $R/gpurun_out/prof/n128* $R/gpurun_out/pmc/*
  cd /tmp
  BARGS="--cpu-sample 0 --exact-extra 0 --n128-extra 0 --boot-extra 0 --shard-extra 0 --group-extra 0"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 1 $BARGS > $R/gpurun_out/prof/bench.json 2> $R/gpurun_out/prof/bench.err; echo "stats bench rc=$?"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o n128 -- python $R/scripts/n128_run.py 6 > $R/gpurun_out/prof/n128.json 2> $R/gpurun_out/prof/n128.err; echo "stats n128 rc=$?"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o bench_$C -- python $R/bench.py --steps 2 --warmup 1 $BARGS > $R/gpurun_out/pmc/bench_$C.json 2> $R/gpurun_out/pmc/bench_$C.err; echo "pmc bench $C rc=$?"
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o n128_$C -- python $R/scripts/n128_run.py 2 > $R/gpurun_out/pmc/n128_$C.json 2> $R/gpurun_out/pmc/n128_$C.err; echo "pmc n128 $C rc=$?"
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o calib_$C -- python -c "
import sys; sys.path.insert(0, '$R')
from psmc_amd import hip
print(hip.stream_probe(1 << 27))" > $R/gpurun_out/pmc/calib_$C.out 2> $R/gpurun_out/pmc/calib_$C.err; echo "calib $C rc=$?"
  done
  SQC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"
  echo "rocprofv3 --kernel-trace --pmc $SQC -- python scripts/n128_run.py 2  (scripts/lease.sh prof)" > $R/gpurun_out/pmc/n128_SQ.cmd
  timeout 600 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $R/gpurun_out/pmc -o n128_SQ -- python $R/scripts/n128_run.py 2 > $R/gpurun_out/pmc/n128_SQ.json 2> $R/gpurun_out/pmc/n128_SQ.err; echo "sq n128 rc=$?"
  echo "rocprofv3 --kernel-trace --pmc $SQC -- python bench.py --steps 2 --warmup 1 $BARGS  (scripts/lease.sh prof)" > $R/gpurun_out/pmc/n64_SQ.cmd
  timeout 600 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $R/gpurun_out/pmc -o n64_SQ -- python $R/bench.py --steps 2 --warmup 1 $BARGS > $R/gpurun_out/pmc/n64_SQ.json 2> $R/gpurun_out/pmc/n64_SQ.err; echo "sq n64 rc=$?"
  cd $R
  python scripts/prof_summary.py lease 30000001 | tail -30
  PROF_NAME=n128 PROF_STATES=128 PROF_CMD="python scripts/n128_run.py" python scripts/prof_summary.py lease 30000001 | tail -20
  python scripts/sq_summary.py lease n64 | head -12; python scripts/sq_summary.py lease n128 | head -10
  mkdir -p gpurun_out/summaries; cp profiles/lease_* profiles/pmc_traffic*.json gpurun_out/summaries/ 2>/dev/null   # only gpurun_out/ travels back
  find gpurun_out/pmc gpurun_out/prof -name "*.csv" -size +3M -delete
}
case "$task" in
  suite) do_suite "$@" ;;
  bench) do_bench "$@" ;;
  sweep)
    CF=(); for c in "$@"; do CF+=(--cfg "$c"); done
    timeout 1200 python scripts/shard_sweep.py "${CF[@]}" --shares "${SHARES-8,4,2,1}" --chr "${CHR-500000}" --repeat "${REPEAT-1}" --factored "${FACTORED-0}" --n-states "${STATES-64}" --warmup "${WARMUP-10}" --steps "${STEPS-12}" --out gpurun_out/sweep.json > gpurun_out/sweep.log 2> gpurun_out/sweep.err
    echo "sweep rc=$?"; tail -3 gpurun_out/sweep.err | cut -c1-300; show_sweep gpurun_out/sweep.json ;;
  timeline)
    cd /tmp; rm -rf $R/gpurun_out/prof/tl*
    timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/scripts/shard_sweep.py --shares "${1-8}" --chr "${2-0}" --cfg "${CFG-}" --factored "${FACTORED-0}" --steps "${STEPS-4}" --warmup "${WARMUP-8}" > $R/gpurun_out/tl.log 2>&1; echo "rocprof rc=$?"
    cd $R; python scripts/prof_timeline.py $(ls -t gpurun_out/prof/tl*.db | head -1) ${LASTK-k_reduce2} all | tee "gpurun_out/timeline${TAG-}.txt" | cut -c1-120
    python scripts/timeline_steps.py $(ls -t gpurun_out/prof/tl*.db | head -1) ${LASTK-k_reduce2} 2 | tee "gpurun_out/timeline_steps${TAG-}.txt" ;;
  prof) do_prof ;;
  wtrace)
    make -s -C psmc_amd/csrc clean; make -s -C psmc_amd/csrc EXTRA=-DPSMC_TRACE_SWEEP 2>&1 | grep -E "error" ; TRACE_STEPS="${TRACE_STEPS-0}" timeout 300 python scripts/sweep_trace.py "${1-1}" "${2-}" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wave_trace.txt | cut -c1-230
    make -s -C psmc_amd/csrc clean; make -s -C psmc_amd/csrc 2>&1 | grep -E "error" ;;
  probes) timeout 300 python scripts/probes.py 2>&1 | grep -v amdgpu.ids ;;
  trace) timeout 600 python scripts/estep_trace.py "${1-1}" "${2-}" 2>&1 | grep -v amdgpu.ids ;;
  northstar) NS_SEPARATE="${NS_SEPARATE-1}" timeout 2400 python scripts/northstar.py gpurun_out/northstar.json > gpurun_out/ns.log 2> gpurun_out/ns.err; echo "northstar rc=$?"; tail -5 gpurun_out/ns.err | cut -c1-300; tail -40 gpurun_out/ns.log | cut -c1-300 ;;
  final) do_suite; do_bench; do_prof ;;
  *) echo "unknown task $task"; exit 2 ;;
esac
