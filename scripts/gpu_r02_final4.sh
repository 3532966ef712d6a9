#!/bin/bash
# round 2, last GPU call: the final build -- whole GPU suite, EM parity over 25 rounds, the bench line as the driver runs it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/f4_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/f4_pytest.log | head -1 | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/f4_pytest.log | head -6 | cut -c1-250
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f4_bench.json 2> gpurun_out/f4_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/f4_bench.json')); print(r['ms_per_step'], r['value'], r['steady_state']['ms_per_step'], r['roofline']['frac'], r['roofline']['kernel'], r['factored_stats']['ms_per_step'], r['exact_mode']['ms_per_step'], r['n128']['ms_per_step'], r['n128']['factored_stats']['ms_per_step'])"
timeout 300 python scripts/em_parity.py gpurun_out/f4_em_parity.json gpurun_out/f4_traj_n64.json > gpurun_out/f4_em_parity.out 2> gpurun_out/f4_em_parity.err
echo "em_parity rc=$?"; tail -6 gpurun_out/f4_em_parity.err | cut -c1-250
