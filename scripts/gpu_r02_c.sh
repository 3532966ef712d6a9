#!/bin/bash
# round 2, GPU call C: factored determinism probe, the tests touched since call B, bench with the n=128 factored extra
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/dbg_factored_determinism.py > gpurun_out/c_dbg.log 2>&1
echo "dbg rc=$?"; cat gpurun_out/c_dbg.log | cut -c1-230
timeout 600 python -m pytest tests -m gpu -q --maxfail=30 -k "batch or n128 or wide or errors or config5 or boot or fast_mode" > gpurun_out/c_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/c_pytest.log | cut -c1-250
timeout 300 python bench.py --cpu-sample 0 --exact-extra 0 > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/c_bench.json')); print(r['ms_per_step'], r['steady_state']['ms_per_step'], json.dumps(r['n128'])[:1500])"
