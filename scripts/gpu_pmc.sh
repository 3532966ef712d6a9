#!/bin/bash
# HBM traffic of the E-step kernels from the rocprofv3 PMC counters: separate passes for
# FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: TCC slots), each with --kernel-trace only,
# plus the same two passes over a copy kernel of known size to calibrate the counters.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
python -c "import __graft_entry__ as g; g.build()" > $R/gpurun_out/build.log 2>&1 || { tail -30 $R/gpurun_out/build.log; exit 1; }
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o bench_$C -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 0 > $R/gpurun_out/pmc/bench_$C.json 2> $R/gpurun_out/pmc/bench_$C.err
  echo "bench $C exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o calib_$C -- python -c "
import sys; sys.path.insert(0, '$R')
from psmc_amd import hip
print(hip.stream_probe(1 << 27))" > $R/gpurun_out/pmc/calib_$C.out 2> $R/gpurun_out/pmc/calib_$C.err
  echo "calib $C exit $?"; cat $R/gpurun_out/pmc/calib_$C.out
done
cd $R; ls gpurun_out/pmc | head -30
python scripts/pmc_summary.py gpurun_out/pmc | tee gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete
