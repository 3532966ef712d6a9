#!/bin/bash
# round 2, GPU call U: eight tiles per wave in the bulk sweeps (lanes8) on the final build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "lanes8=1" "lanes8=0" "lanes8=1 warmup=2560" ; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/u_bench_$tag.json 2> gpurun_out/u_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/u_bench_$tag.json"))
k=r["roofline"]["kernels_ms"]; fk=r["factored_stats"].get("kernels_ms") or {}
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  fwd_sweep %.2f expect %.2f | factored fwd %.2f acc %.2f | repairs %s" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], k["fwd_sweep"], k["expect"], fk.get("fwd_sweep",0), fk.get("expect",0), r["config"].get("repair_rounds")))
PY
done
