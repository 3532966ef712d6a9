#!/bin/bash
# rocprofv3 kernel trace of a few fast-mode E-steps + timeline of the last one
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof/tl*
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/bench.py --steps 4 --warmup 3 --cpu-sample 0 --exact-extra 0 ${BENCH_ARGS:-} > $R/gpurun_out/prof/tl.json 2> $R/gpurun_out/prof/tl.err
echo "rocprof exit $?"; tail -1 $R/gpurun_out/prof/tl.json | cut -c1-300
cd $R
python scripts/prof_timeline.py $(ls gpurun_out/prof/tl*.db | tail -1)
