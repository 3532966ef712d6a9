#!/bin/bash
# round 3, GPU call B: pipe probe 2 + placement probe; small-input option sweep (kc_warm, kc_min, two_phase, warm_shift)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/r03/probes.py > gpurun_out/b_probes.log 2>&1; echo "probes rc=$?"; cat gpurun_out/b_probes.log | grep -v amdgpu.ids
CFGS=()
for ch in 256 512; do for kw in 0 1; do for km in 4 2; do for tp in 2 0; do
  CFGS+=(--cfg "chunk=$ch two_phase=$tp warm_shift=0 kc_warm=$kw kc_min=$km")
done; done; done; done
CFGS+=(--cfg "chunk=256 two_phase=0 warm_shift=0 kc_warm=1 kc_min=2 warmup=2048" --cfg "chunk=256 two_phase=0 warm_shift=0 kc_warm=1 kc_min=2 warmup=4096" --cfg "chunk=256 two_phase=0 warm_shift=0 kc_min=0" --cfg "chunk=256 two_phase=0 warm_shift=0 learn=0")
timeout 900 python scripts/shard_sweep.py "${CFGS[@]}" --shares 8 --out gpurun_out/b_sweep.json > gpurun_out/b_sweep.log 2> gpurun_out/b_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/b_sweep.err | cut -c1-300
python - <<'PY'
import json
for r in json.load(open("gpurun_out/b_sweep.json")):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-62s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
