#!/bin/bash
# round 2, GPU call I: the bench line after the make_line refactor (driver's command), new group test,
# psmc_boot fast mode with two contexts on one GPU (PSMC_HIP_DEVICES=0,0: does a second driver thread fill the gaps?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/i_bench.json')); print(r['ms_per_step'], r['steady_state']['ms_per_step'], json.dumps({k:v for k,v in r['roofline'].items() if k in ('bound','kernel','achieved','peak','frac','traffic','kernel_ms','also')})[:1200])"
timeout 300 python -m pytest tests -m gpu -q -k "group or two_ranks" > gpurun_out/i_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/i_pytest.log | cut -c1-200
BOOT_ITERS=4 BOOT_FAST_ONLY=1 PSMC_HIP_DEVICES=0,0 timeout 300 python scripts/time_boot.py gpurun_out/i_boot_timing_2ctx.json > gpurun_out/i_boot.out 2> gpurun_out/i_boot.err
echo "boot rc=$?"; tail -3 gpurun_out/i_boot.err | cut -c1-250
