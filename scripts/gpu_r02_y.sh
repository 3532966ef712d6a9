#!/bin/bash
# round 2, GPU call Y: warm-up-only walks for the heads of transfer-matrix chains -- parity under varying poison, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or factored or config3 or batch or learn" > gpurun_out/y_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/y_pytest.log | cut -c1-250; grep -n "^E  " gpurun_out/y_pytest.log | head -8 | cut -c1-250
for cfg in "walk_heads=0" "walk_heads=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/y_bench_$tag.json 2> gpurun_out/y_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/y_bench_$tag.json"))
k=r["roofline"]["kernels_ms"]; fk=r["factored_stats"].get("kernels_ms") or {}
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  fwd_sweep %.2f expect %.2f | factored fwd %.2f acc %.2f total %.2f | repairs %s" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], k["fwd_sweep"], k["expect"], fk.get("fwd_sweep",0), fk.get("expect",0), fk.get("total",0), r["config"].get("repair_rounds")))
PY
done
BENCH_ARGS="--n128-extra 0" bash scripts/gpu_timeline.sh > gpurun_out/y_timeline.log 2>&1; tail -18 gpurun_out/y_timeline.log | cut -c1-120
