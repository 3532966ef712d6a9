#!/bin/bash
# round 2, GPU call L: sub-tiled column kernel -- parity under varying poison, A/B timing, timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or config3 or batch" > gpurun_out/l_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/l_pytest.log | cut -c1-250
for cfg in "kc_sub=4" "kcol_impl=0" "kc_sub=2" "kc_sub=8" "kc_sub=4 kc_div=8" "kcol_impl=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/l_bench_$tag.json 2> gpurun_out/l_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/l_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  fwd_sweep %.2f fused %.2f" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["roofline"]["kernels_ms"]["fwd_sweep"], r["roofline"]["kernels_ms"]["expect"]))
PY
done
BENCH_ARGS="--n128-extra 0" bash scripts/gpu_timeline.sh > gpurun_out/l_timeline.log 2>&1; tail -16 gpurun_out/l_timeline.log | cut -c1-120
