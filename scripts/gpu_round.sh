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
# the N>1 code path of bench.py (2 ranks on the one GPU of this box, gloo instead of RCCL)
BENCH_SINGLE_GPU_TEST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --bins 2000000 --segments 30 > gpurun_out/bench_2rank_test.json 2> gpurun_out/bench_2rank_test.err
echo "2-rank bench path exit $?"; tail -2 gpurun_out/bench_2rank_test.err | cut -c1-200; cut -c1-300 gpurun_out/bench_2rank_test.json
timeout 900 python bench.py --steps 5 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench_30m.json 2> gpurun_out/bench_30m.err
echo "bench30m exit $?"; tail -3 gpurun_out/bench_30m.err; cat gpurun_out/bench_30m.json
