#!/bin/bash
# One GPU-box session: build, primitive self-test, parity tests, bench.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
if [ "${SWEEP:-0}" = "1" ]; then
  timeout 900 python scripts/gpu_sweep.py > gpurun_out/sweep.log 2> gpurun_out/sweep.err; echo "sweep exit $?"; cat gpurun_out/sweep.log; tail -3 gpurun_out/sweep.err
fi
timeout 900 python bench.py --steps 5 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench_30m.json 2> gpurun_out/bench_30m.err
echo "bench30m exit $?"; tail -3 gpurun_out/bench_30m.err; cat gpurun_out/bench_30m.json
