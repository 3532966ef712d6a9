#!/bin/bash
# round 3, GPU call K: warm-up length / tile length re-tuned now that the first launch of the back half no longer waits for the runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]; pl = r["plan"]
    print("%-16s %-26s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f cnt %.2f  warm f %4.0f b %4.0f max %d/%d glued %d/%d" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["expect"],
          pl["warm_fwd_mean"], pl["warm_bwd_mean"], pl["warm_fwd_max"], pl["warm_bwd_max"], pl["glued_fwd"], pl["glued_bwd"]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "warmup=2048" --cfg "warmup=2560" --cfg "warmup=3584" --cfg "warmup=2560 warm_shift=2" --cfg "kc_min=2" --cfg "kc_min=8" --cfg "kcol_prio=1" --cfg "group_cap=65536" --shares 1 --chr 0 --warmup 10 --steps 14 --out gpurun_out/k_sweep.json > gpurun_out/k_sweep.log 2> gpurun_out/k_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/k_sweep.err | cut -c1-300; show gpurun_out/k_sweep.json
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --cfg "warmup=2560" --cfg "warmup=2048" --shares 1 --chr 0 --warmup 10 --steps 14 --out gpurun_out/k_sweep_fac.json > gpurun_out/k_sweep_fac.log 2> gpurun_out/k_sweep_fac.err
echo "sweep fac rc=$?"; show gpurun_out/k_sweep_fac.json
