#!/bin/bash
# round 3, GPU call I: adaptive per-tile warm-ups: parity suite, A/B against adapt=0 on every shard size (full counts and factored)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/i_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/i_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/i_pytest.log | head -12 | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]; pl = r["plan"]
    print("%-16s %-22s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f cnt %.2f  warm f %4.0f b %4.0f max %d/%d glued %d/%d" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["expect"],
          pl["warm_fwd_mean"], pl["warm_bwd_mean"], pl["warm_fwd_max"], pl["warm_bwd_max"], pl["glued_fwd"], pl["glued_bwd"]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "adapt=0" --shares 8,4,2,1 --warmup 20 --steps 12 --out gpurun_out/i_sweep.json > gpurun_out/i_sweep.log 2> gpurun_out/i_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/i_sweep.err | cut -c1-300; show gpurun_out/i_sweep.json
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --cfg "adapt=0" --shares 8,1 --warmup 20 --steps 12 --out gpurun_out/i_sweep_fac.json > gpurun_out/i_sweep_fac.log 2> gpurun_out/i_sweep_fac.err
echo "sweep factored rc=$?"; tail -3 gpurun_out/i_sweep_fac.err | cut -c1-300; show gpurun_out/i_sweep_fac.json
timeout 600 python bench.py --engine group --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/i_bench_group1.json 2> gpurun_out/i_bench_group1.err; echo "group1 rc=$?"; cut -c1-330 gpurun_out/i_bench_group1.json
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 --boot-extra 0 --shard-extra 0 > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/i_bench.json").read().strip().splitlines()[-1])
print("headline %.3f ms  %.3e bins/s  frac %.3f  steady %s  factored %s" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r.get("steady_state", {}).get("ms_per_step"), r.get("factored_stats", {}).get("ms_per_step")))
print("kernels", {k: round(v, 2) for k, v in r["roofline"]["kernels_ms"].items()}); print("group", r.get("group_engine"))
PY
