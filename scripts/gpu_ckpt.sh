#!/bin/bash
# checkpointed factored E-step: parity subset, timing against ckpt=0, HBM probes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_estep.py -m gpu -q --no-header -p no:cacheprovider -k "factored or fused" -x > gpurun_out/ckpt_pytest.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/ckpt_pytest.log
timeout 300 python scripts/time_factored.py > gpurun_out/ckpt_time1.log 2>&1; echo "ckpt=1"; grep -v "^full" gpurun_out/ckpt_time1.log | head -8
timeout 300 python scripts/time_factored.py ckpt=0 > gpurun_out/ckpt_time0.log 2>&1; echo "ckpt=0"; grep -v "^full" gpurun_out/ckpt_time0.log | head -8
timeout 200 python -c "
from psmc_amd import hip
print(hip.hbm_probe(8 << 30))
print(hip.hbm_probe(16 << 30))" > gpurun_out/hbm_probe.log 2>&1; cat gpurun_out/hbm_probe.log
