#!/bin/bash
# round 3, GPU call Q: optimistic tail on the idle counts stream: parity (plain + varying poison), cost of an E-step that needs repairs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_estep.py -m gpu -q --no-header -p no:cacheprovider -k "fast" > gpurun_out/q_pytest.log 2>&1
echo "pytest fast rc=$?"; tail -3 gpurun_out/q_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/q_pytest.log | head -8 | cut -c1-250
PSMC_HIP_POISON=vary timeout 900 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --no-header -p no:cacheprovider -k "fast or shard or config3" > gpurun_out/q_pytest_poison.log 2>&1
echo "pytest poison rc=$?"; tail -3 gpurun_out/q_pytest_poison.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/q_pytest_poison.log | head -8 | cut -c1-250
python scripts/r03/adapt_trace.py 1 2>&1 | grep -v amdgpu | sed -n '1,4p;18,30p'
