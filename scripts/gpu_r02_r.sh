#!/bin/bash
# round 2, GPU call R: after the dependency fix of the factored run tiles -- the fast suite twice under varying poison, once without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --maxfail=30 -k "fast or factored or config3 or config5 or n128 or batch or learn" > gpurun_out/r_pytest_$i.log 2>&1
echo "pytest $i rc=$?"; tail -4 gpurun_out/r_pytest_$i.log | cut -c1-250; grep -n "^E  .*AssertionError\|^E   *assert [0-9]" gpurun_out/r_pytest_$i.log | head
done
timeout 600 python -m pytest tests/test_gpu_estep.py -m gpu -q --maxfail=30 -k "fused_backward_counts or factored" > gpurun_out/r_pytest_3.log 2>&1
echo "pytest 3 (no poison) rc=$?"; tail -3 gpurun_out/r_pytest_3.log | cut -c1-250
