#!/bin/bash
# round 2, GPU call O: 128-state column kernel (k_kcol2_struct<4>) -- parity under varying poison, config-5 timing A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests -m gpu -q --maxfail=30 -k "n128 or config5 or wide or fast_learns or odd_tilings or factored or many_small" > gpurun_out/o_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/o_pytest.log | cut -c1-250
for cfg in "kcol_impl=1" "kcol_impl=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 6 --warmup 5 --cpu-sample 0 --exact-extra 0 $opts > gpurun_out/o_bench_$tag.json 2> gpurun_out/o_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/o_bench_$tag.json")); n=r["n128"]
print("   n64 moving %.2f | n128 full %.2f  factored %.2f  fwd_sweep %.2f / %.2f" % (r["ms_per_step"], n["ms_per_step"], n["factored_stats"]["ms_per_step"], n["kernels_ms"]["fwd_sweep"], n["factored_stats"]["kernels_ms"]["fwd_sweep"]))
PY
done
