#!/bin/bash
# round 2, GPU call K: transfer matrices with one column per lane (kcol_impl=1) -- parity under varying poison, A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or config3 or group or batch" > gpurun_out/k_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/k_pytest.log | cut -c1-250
for cfg in "kcol_impl=1" "kcol_impl=0" "kcol_impl=1 kc_min=3" "kcol_impl=1 kc_div=8"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/k_bench_$tag.json 2> gpurun_out/k_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/k_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  items %s repairs %s/%s fwd_sweep %.2f fused %.2f" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["config"]["sweep_items"], r["config"]["repair_rounds"], r["config"]["repaired_tiles"], r["roofline"]["kernels_ms"]["fwd_sweep"], r["roofline"]["kernels_ms"]["expect"]))
PY
done
BENCH_ARGS="--n128-extra 0" bash scripts/gpu_timeline.sh > gpurun_out/k_timeline.log 2>&1; tail -16 gpurun_out/k_timeline.log | cut -c1-120
