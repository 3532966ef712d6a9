#!/bin/bash
# round 2, GPU call P: forward-scaled back halves (fused: + interleaved matrix instructions) -- parity under varying poison, A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or factored or config3 or config5 or n128 or batch or learn" > gpurun_out/p_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/p_pytest.log | cut -c1-250
for cfg in "count_impl=2" "count_impl=1" "count_impl=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/p_bench_$tag.json 2> gpurun_out/p_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/p_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  kernels %s" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["roofline"]["kernels_ms"]))
print("   factored kernels", r["factored_stats"].get("kernels_ms"))
PY
done
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --exact-extra 0 --n128-extra 1 > gpurun_out/p_bench_n128.json 2> gpurun_out/p_bench_n128.err
python - <<PY
import json
r=json.load(open("gpurun_out/p_bench_n128.json"))
print("n128", json.dumps(r.get("n128"))[:600])
PY
