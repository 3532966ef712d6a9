#!/bin/bash
# round 3, GPU call E: walks outside the merged grid, kc_sub rule, eight-tiles-per-wave backward warm-up
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-40s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "merge1=2" --cfg "merge1=0" --cfg "lanes8=2" --cfg "kc_sub=1" --cfg "kc_sub=4" --cfg "walk_impl=0" --cfg "lanes8=2 walk_impl=0" \
   --shares 8,4,2 --out gpurun_out/e_sweep.json > gpurun_out/e_sweep.log 2> gpurun_out/e_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/e_sweep.err | cut -c1-300; show gpurun_out/e_sweep.json
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "lanes8=2" --cfg "lanes8=2 merge1=1" --cfg "merge1=1" \
   --shares 1 --chr 0 --out gpurun_out/e_sweep_full.json > gpurun_out/e_sweep_full.log 2> gpurun_out/e_sweep_full.err
echo "sweep full rc=$?"; tail -3 gpurun_out/e_sweep_full.err | cut -c1-300; show gpurun_out/e_sweep_full.json
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --cfg "lanes8=1" --cfg "lanes8=2" --cfg "lanes8=2 merge1=1" --cfg "merge1=1" \
   --shares 1,8 --chr 0 --out gpurun_out/e_sweep_fac.json > gpurun_out/e_sweep_fac.log 2> gpurun_out/e_sweep_fac.err
echo "sweep factored rc=$?"; tail -3 gpurun_out/e_sweep_fac.err | cut -c1-300; show gpurun_out/e_sweep_fac.json
