#!/bin/bash
# round 2, GPU call F: option A/B on the new defaults (lanes8, two_phase, tile count), then the whole GPU test-suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "learn=1" "lanes8=1" "two_phase=1" "struct_tiles=6144" "struct_tiles=4096" "struct_tiles=12288" "kc_min=8" "kc_div=32"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 25 --warmup 25 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/f_bench_$tag.json 2> gpurun_out/f_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/f_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  first %.1f  tiles %s items %s repairs %s/%s fwd_sweep %.2f fused %.2f" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["first_call_ms"], r["config"]["tiles"], r["config"]["sweep_items"], r["config"]["repair_rounds"], r["config"]["repaired_tiles"], r["roofline"]["kernels_ms"]["fwd_sweep"], r["roofline"]["kernels_ms"]["expect"]))
PY
done
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/f_pytest.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/f_pytest.log | cut -c1-250
