#!/bin/bash
# round 2, GPU call D: who reads memory nobody wrote?  varying-garbage poison + option variants; then the whole
# GPU test-suite under NaN poison (any dependency on uninitialised memory shows up as NaN / mismatch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 400 python scripts/dbg_factored_determinism.py > gpurun_out/d_dbg.log 2>&1
echo "dbg rc=$?"; grep -E "==|False" gpurun_out/d_dbg.log | cut -c1-200 | head -80
PSMC_HIP_POISON=1 timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/d_pytest_poison.log 2>&1
echo "pytest(poison) rc=$?"; tail -30 gpurun_out/d_pytest_poison.log | cut -c1-250
