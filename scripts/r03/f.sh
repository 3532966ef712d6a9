#!/bin/bash
# round 3, GPU call F: after the prune + single-stream chain + optimistic tail: parity suite (also under varying poison for the fast tests), sweep, timelines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > gpurun_out/f_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/f_pytest.log | cut -c1-250; grep -n "^E  " gpurun_out/f_pytest.log | head -8 | cut -c1-250
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py -m gpu -q -x --no-header -p no:cacheprovider -k "fast" > gpurun_out/f_pytest_poison.log 2>&1
echo "pytest (poison=vary, fast) rc=$?"; tail -3 gpurun_out/f_pytest_poison.log | cut -c1-250; grep -n "^E  " gpurun_out/f_pytest_poison.log | head -8 | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-40s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "merge1=0" --shares 8,4,2,1 --out gpurun_out/f_sweep.json > gpurun_out/f_sweep.log 2> gpurun_out/f_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/f_sweep.err | cut -c1-300; show gpurun_out/f_sweep.json
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --shares 8,1 --out gpurun_out/f_sweep_fac.json > gpurun_out/f_sweep_fac.log 2> gpurun_out/f_sweep_fac.err
echo "sweep factored rc=$?"; tail -3 gpurun_out/f_sweep_fac.err | cut -c1-300; show gpurun_out/f_sweep_fac.json
bash scripts/r03/d.sh 2>&1 | cut -c1-130
