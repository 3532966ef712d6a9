#!/bin/bash
# round 2, GPU call Z: the last build -- whole GPU suite, the bench line as the driver runs it, rocprofv3 kernel stats of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/z_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/z_pytest.log | cut -c1-250
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/z_bench.json')); print(r['ms_per_step'], r['value'], r['steady_state']['ms_per_step'], r['roofline']['frac'], r['roofline']['kernel'], r['factored_stats']['ms_per_step'], r['exact_mode']['ms_per_step'], r['n128']['ms_per_step'], r['n128']['factored_stats']['ms_per_step'])"
rm -rf gpurun_out/prof gpurun_out/pmc
timeout 300 bash scripts/gpu_prof.sh > gpurun_out/z_prof.log 2>&1
echo "prof rc=$?"
