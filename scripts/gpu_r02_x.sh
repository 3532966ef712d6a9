#!/bin/bash
# round 2, GPU call X: factored path options on the final build (ckpt, two_phase)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "ckpt=1" "ckpt=0" "two_phase=1" "two_phase=1 ckpt=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 10 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 0 $opts > gpurun_out/x_bench_$tag.json 2> gpurun_out/x_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/x_bench_$tag.json"))
k=r["roofline"]["kernels_ms"]; fk=r["factored_stats"].get("kernels_ms") or {}
print("   moving %.2f ms  factored %.2f ms | factored kernels %s" % (r["ms_per_step"], r["factored_stats"]["ms_per_step"], {a: round(b, 2) for a, b in fk.items()}))
PY
done
