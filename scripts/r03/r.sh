#!/bin/bash
# round 3, GPU call R: multi-wave chain kernel with prefetch: parity (plain + poison), shard sizes, config 5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --no-header -p no:cacheprovider -k "fast or shard or config3 or config5 or n128" > gpurun_out/r_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/r_pytest.log | head -8 | cut -c1-250
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py -m gpu -q --no-header -p no:cacheprovider -k "fast" > gpurun_out/r_pytest_poison.log 2>&1
echo "pytest poison rc=$?"; tail -3 gpurun_out/r_pytest_poison.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/r_pytest_poison.log | head -8 | cut -c1-250
timeout 900 python scripts/shard_sweep.py --cfg "" --repeat 2 --shares 8,4,1 --warmup 10 --steps 12 --out gpurun_out/r_sweep.json > gpurun_out/r_sweep.log 2> gpurun_out/r_sweep.err
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r_sweep.json")):
    if "error" in r: print(r["workload"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]; print("%-16s %7.3f ms (min %6.3f) tot %.2f fwd %.2f cnt %.2f" % (r["workload"], r["ms_median"], r["ms_min"], k["total"], k["fwd_sweep"], k["expect"]))
PY
timeout 900 python scripts/shard_sweep.py --factored 1 --cfg "" --shares 1 --chr 0 --warmup 10 --steps 12 --out gpurun_out/r_sweep_fac.json > /dev/null 2> gpurun_out/r_sweep_fac.err
python -c "
import json
for r in json.load(open('gpurun_out/r_sweep_fac.json')): print('factored', r['workload'], round(r['ms_median'],3), r['kernels_ms'])"
cd /tmp; rm -rf $R/gpurun_out/prof/r_tl*
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o r_tl -- python $R/scripts/shard_sweep.py --shares '' --chr 500000 --steps 4 --warmup 8 > $R/gpurun_out/r_tl.log 2>&1
cd $R; python scripts/prof_timeline.py $(ls gpurun_out/prof/r_tl*.db | tail -1) k_reduce2 all 2>&1 | grep -E "kchain|walk|sweep|count|fwd_struct1|reduce2" | cut -c1-110
timeout 600 python scripts/r03/n128_run.py 8 2>&1 | tail -1 | cut -c1-400
