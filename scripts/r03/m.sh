#!/bin/bash
# round 3, GPU call M: wave priorities of phase 1 re-tuned for the two-launch plans (runs late, forward sweep split)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CFGS=()
for sf in 0 1; do for bp in 0 1; do for wp in 3 1; do for kp in 2 0; do CFGS+=(--cfg "split_fwd=$sf bwd_prio=$bp walk_prio=$wp kcol_prio=$kp"); done; done; done; done
timeout 900 python scripts/shard_sweep.py "${CFGS[@]}" --repeat 2 --shares 1 --chr 0 --warmup 8 --steps 12 --out gpurun_out/m_sweep.json > gpurun_out/m_sweep.log 2> gpurun_out/m_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/m_sweep.err | cut -c1-300
python - <<'PY'
import json, collections
agg = collections.defaultdict(list); k2 = {}
for r in json.load(open("gpurun_out/m_sweep.json")):
    if "error" in r: print(r["cfg"], "ERROR", r["error"][:100]); continue
    agg[r["cfg"]].append(r["ms_median"]); k2[r["cfg"]] = r["kernels_ms"]
for k, v in sorted(agg.items(), key=lambda x: sum(x[1]) / len(x[1])): print("%-60s mean %.3f  %s  fwd %.2f cnt %.2f" % (k, sum(v) / len(v), [round(x, 2) for x in v], k2[k]["fwd_sweep"], k2[k]["expect"]))
PY
