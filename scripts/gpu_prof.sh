#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (per-kernel average durations).
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 0 ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/prof/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/bench.err
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && column -s, -t "$f" | cut -c1-220 | head -30
tail -2 gpurun_out/prof/bench.err
# keep only the summaries (the raw trace can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
