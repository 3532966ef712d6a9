#!/bin/bash
# round 2, GPU call AA: bt with its own scaling again (weights follow both scale factors) -- parity under varying poison, stress of tiny tiles, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or factored or config3 or config5 or n128 or batch or learn" > gpurun_out/aa_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/aa_pytest.log | head -1 | cut -c1-250; grep -n "^E  " gpurun_out/aa_pytest.log | head -6 | cut -c1-250
DBG_FACTORED=1 timeout 300 python scripts/dbg_flaky_tiling.py 120 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --exact-extra 0 --n128-extra 1 > gpurun_out/aa_bench.json 2> gpurun_out/aa_bench.err
echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("gpurun_out/aa_bench.json"))
k=r["roofline"]["kernels_ms"]; fk=r["factored_stats"].get("kernels_ms") or {}
print("   moving %.2f ms  steady %.2f ms  factored %.2f ms  fwd_sweep %.2f expect %.2f | factored fwd %.2f acc %.2f | n128 %.2f / %.2f" % (r["ms_per_step"], r["steady_state"]["ms_per_step"], r["factored_stats"]["ms_per_step"], k["fwd_sweep"], k["expect"], fk.get("fwd_sweep",0), fk.get("expect",0), r["n128"]["ms_per_step"], r["n128"]["factored_stats"]["ms_per_step"]))
PY
