#!/bin/bash
# round 2, GPU call M: wave priority of the column kernel; two_phase=1 again on the new balance
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "kcol_prio=2" "kcol_prio=1" "kcol_prio=0" "kcol_prio=1 kc_sub=2" "two_phase=1" "warmup=2560"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/m_bench_$tag.json 2> gpurun_out/m_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/m_bench_$tag.json"))
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  fwd_sweep %.2f fused %.2f items %s" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], r["roofline"]["kernels_ms"]["fwd_sweep"], r["roofline"]["kernels_ms"]["expect"], r["config"]["sweep_items"]))
PY
done
