#!/bin/bash
# One GPU-box session: build, primitive self-test, parity tests, bench.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6
timeout 120 python -c "
from psmc_amd import hip
print('selftest mask:', hip.selftest(0))
" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 --bins 3000000 --segments 40 --cpu-sample 300000 > gpurun_out/bench_3m.json 2> gpurun_out/bench_3m.err
echo "bench3m exit $?"; tail -3 gpurun_out/bench_3m.err; cat gpurun_out/bench_3m.json
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_30m.json 2> gpurun_out/bench_30m.err
echo "bench30m exit $?"; tail -3 gpurun_out/bench_30m.err; cat gpurun_out/bench_30m.json
