#!/bin/bash
# round 2, GPU call A: EM parity (-N25, four configurations, config 2 + config 3), the new bench line, warm-up sweep,
# cross-wave pipe probe, then the whole GPU test-suite.  Everything under its own timeout; logs in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python scripts/em_parity.py gpurun_out/r02_em_parity.json gpurun_out/traj_n64.json > gpurun_out/a_em_parity.out 2> gpurun_out/a_em_parity.err
echo "em_parity rc=$?"
timeout 300 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/a_bench.json
timeout 60 python -c "
import json
from psmc_amd import hip
print(json.dumps(dict(pipe_probe=hip.pipe_probe(), microbench=hip.microbench()), indent=1))" > gpurun_out/a_pipe_probe.json 2> gpurun_out/a_pipe_probe.err
echo "pipe rc=$?"
for w in 1024 2048 3072; do
  timeout 150 python bench.py --opt warmup=$w --exact-extra 0 --n128-extra 0 --cpu-sample 0 > gpurun_out/a_bench_w$w.json 2> gpurun_out/a_bench_w$w.err
  echo "warmup $w rc=$?"
done
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/a_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/a_pytest.log
