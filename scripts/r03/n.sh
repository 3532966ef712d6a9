#!/bin/bash
# round 3, GPU call N: warm-up length on shard-sized inputs (one-round plan)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "warmup=1536" --cfg "warmup=2048" --cfg "warmup=2560" --cfg "warmup=2048 kc_min=2" --cfg "kc_min=2" --cfg "kc_min=8" --repeat 2 --shares 8 --warmup 10 --steps 12 --out gpurun_out/n_sweep.json > gpurun_out/n_sweep.log 2> gpurun_out/n_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/n_sweep.err | cut -c1-300
python - <<'PY'
import json, collections
agg = collections.defaultdict(list); k2 = {}
for r in json.load(open("gpurun_out/n_sweep.json")):
    if "error" in r: print(r["cfg"], "ERROR", r["error"][:100]); continue
    agg[(r["workload"], r["cfg"])].append(r["ms_median"]); k2[(r["workload"], r["cfg"])] = (r["kernels_ms"], r["plan"])
for k, v in sorted(agg.items(), key=lambda x: (x[0][0], sum(x[1]) / len(x[1]))): print("%-16s %-28s mean %.3f  %s  fwd %.2f cnt %.2f glued %d/%d" % (k[0], k[1], sum(v) / len(v), [round(x, 2) for x in v], k2[k][0]["fwd_sweep"], k2[k][0]["expect"], k2[k][1]["glued_fwd"], k2[k][1]["glued_bwd"]))
PY
