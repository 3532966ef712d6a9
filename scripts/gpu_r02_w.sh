#!/bin/bash
# round 2, GPU call W: fuse128 as the default -- every test that touches 65..128 states, config 5 at full size, the bench extra
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py tests/test_host_cli.py -m gpu -q --maxfail=30 -k "128 or config5 or wide or generic" > gpurun_out/w_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/w_pytest.log | cut -c1-250; grep -n "^E  " gpurun_out/w_pytest.log | head -8 | cut -c1-250
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 1 > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err
echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("gpurun_out/w_bench.json"))["n128"]
print("   n128 full %.2f ms  factored %.2f ms kernels %s" % (r["ms_per_step"], r["factored_stats"]["ms_per_step"], {k: round(v, 2) for k, v in r["kernels_ms"].items()}))
PY
