#!/bin/bash
# round 3, final GPU call: whole suite, the driver's bench command, profiles (rocprofv3 stats + PMC + SQ; bench and config 5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/final_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/final_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/final_pytest.log | head -12 | cut -c1-250
s=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; tail -2 gpurun_out/final_bench.err | cut -c1-300
python - <<'PY'
import json
r = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("headline %.3f ms  %.3e bins/s  frac %.3f  steady %s" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r.get("steady_state", {}).get("ms_per_step")))
print("kernels", {k: round(v, 2) for k, v in r["roofline"]["kernels_ms"].items()})
print("factored", r.get("factored_stats", {}).get("ms_per_step"))
for w in r.get("shard_sweep", {}).get("workloads", []): print("shard", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in w.items() if k != "kernels_ms"})
print("group", r.get("group_engine"))
print("boot", json.dumps(r.get("boot"))[:1200])
n = r.get("n128", {})
print("n128", n.get("ms_per_step"), n.get("ms_min"), n.get("factored_stats", {}).get("ms_per_step") if isinstance(n.get("factored_stats"), dict) else n.get("factored_stats"), n.get("roofline", {}).get("frac"), n.get("config"), n.get("error"))
print("exact", r.get("exact_mode", {}).get("ms_per_step"), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
bash scripts/r03/prof.sh 2>&1 | tail -90 | cut -c1-220
