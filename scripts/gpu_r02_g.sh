#!/bin/bash
# round 2, GPU call G: 128-state transfer-matrix chains (parity tests + config-5 timing), tests touched since call F
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests -m gpu -q --maxfail=30 -k "n128 or wide or learns or config5 or config2 or odd_tilings or many_small" > gpurun_out/g_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/g_pytest.log | cut -c1-250
timeout 300 python bench.py --cpu-sample 0 --exact-extra 0 --steps 10 --warmup 25 > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/g_bench.json')); print(r['ms_per_step'], r['steady_state']['ms_per_step']); print(json.dumps(r['n128'])[:1800])"
timeout 200 python bench.py --cpu-sample 0 --exact-extra 0 --steps 4 --warmup 4 --opt kc_min=0 > gpurun_out/g_bench_nochain.json 2> gpurun_out/g_bench_nochain.err
python -c "
import json; r=json.load(open('gpurun_out/g_bench_nochain.json')); print('kc_min=0:', json.dumps(r['n128'])[:900])"
