#!/bin/bash
# round 3, GPU call O: factored back half -- members of forward runs in a side pass (runs_late for the item-list back half): A/B + parity
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --cfg "runs_late=0" --repeat 3 --shares 1 --chr 0 --warmup 10 --steps 14 --out gpurun_out/o_sweep.json > gpurun_out/o_sweep.log 2> gpurun_out/o_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/o_sweep.err | cut -c1-300
python - <<'PY'
import json, collections
agg = collections.defaultdict(list); k2 = {}
for r in json.load(open("gpurun_out/o_sweep.json")):
    if "error" in r: print(r["cfg"], "ERROR", r["error"][:100]); continue
    agg[(r["workload"], r["cfg"])].append(r["ms_median"]); k2[(r["workload"], r["cfg"])] = r["kernels_ms"]
for k, v in sorted(agg.items(), key=lambda x: (x[0][0], sum(x[1]) / len(x[1]))): print("%-16s %-28s mean %.3f  %s  tot %.2f fwd %.2f cnt %.2f" % (k[0], k[1], sum(v) / len(v), [round(x, 2) for x in v], k2[k]["total"], k2[k]["fwd_sweep"], k2[k]["expect"]))
PY
PSMC_HIP_POISON=vary timeout 1200 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --no-header -p no:cacheprovider -k "factored or config3 or shard or boot or batch" > gpurun_out/o_pytest.log 2>&1
echo "pytest (poison=vary) rc=$?"; tail -4 gpurun_out/o_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/o_pytest.log | head -12 | cut -c1-250
