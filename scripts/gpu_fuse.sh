#!/bin/bash
# fused backward + counts (4 tiles per wave, K = tiles): parity subset and timing against the default path
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_estep.py -m gpu -q --no-header -p no:cacheprovider -k "fused or odd_tilings" -x > gpurun_out/fuse_pytest.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/fuse_pytest.log
timeout 300 python scripts/sweep_factored.py --full "" "fuse=1" ${EXTRA:-} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fuse_time.log
