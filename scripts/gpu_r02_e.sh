#!/bin/bash
# round 2, GPU call E: determinism probe after the dependency fix, test-suite under varying poison, warm_shift A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 300 python scripts/dbg_factored_determinism.py 2 > gpurun_out/e_dbg.log 2>&1
echo "dbg rc=$?"; grep -E "==|False" gpurun_out/e_dbg.log | cut -c1-200 | head -40
for cfg in "warm_shift=2" "warm_shift=0" "warm_shift=2 warmup=3072" "warm_shift=1 warmup=3072" "warm_shift=3 warmup=2048"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 25 --warmup 25 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/e_bench_$tag.json 2> gpurun_out/e_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/e_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  first %.1f  items %s repairs %s/%s fwd_sweep %.2f fused %.2f" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["first_call_ms"], r["config"]["sweep_items"], r["config"]["repair_rounds"], r["config"]["repaired_tiles"], r["roofline"]["kernels_ms"]["fwd_sweep"], r["roofline"]["kernels_ms"]["expect"]))
PY
done
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/e_pytest_poison.log 2>&1
echo "pytest(poison vary) rc=$?"; tail -30 gpurun_out/e_pytest_poison.log | cut -c1-250
