#!/bin/bash
# round 3, final profiles: rocprofv3 kernel stats + PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, calibrated on a
# known copy) of the bench command and of config 5 alone; SQ matrix-pipe counters of the 128-state back halves.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p $R/gpurun_out/prof $R/gpurun_out/pmc
rm -rf $R/gpurun_out/prof/bench* $R/gpurun_out/prof/n128* $R/gpurun_out/pmc/*
export TMPDIR=/tmp
cd /tmp
BARGS="--steps 5 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 0 --boot-extra 0 --shard-extra 0 --group-extra 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py $BARGS > $R/gpurun_out/prof/bench.json 2> $R/gpurun_out/prof/bench.err; echo "stats bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o n128 -- python $R/scripts/r03/n128_run.py 6 > $R/gpurun_out/prof/n128.json 2> $R/gpurun_out/prof/n128.err; echo "stats n128 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o bench_$C -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 0 --boot-extra 0 --shard-extra 0 --group-extra 0 > $R/gpurun_out/pmc/bench_$C.json 2> $R/gpurun_out/pmc/bench_$C.err; echo "pmc bench $C rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o n128_$C -- python $R/scripts/r03/n128_run.py 2 > $R/gpurun_out/pmc/n128_$C.json 2> $R/gpurun_out/pmc/n128_$C.err; echo "pmc n128 $C rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc -o calib_$C -- python -c "
import sys; sys.path.insert(0, '$R')
from psmc_amd import hip
print(hip.stream_probe(1 << 27))" > $R/gpurun_out/pmc/calib_$C.out 2> $R/gpurun_out/pmc/calib_$C.err; echo "calib $C rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc -o n128_SQ -- python $R/scripts/r03/n128_run.py 2 > $R/gpurun_out/pmc/n128_SQ.json 2> $R/gpurun_out/pmc/n128_SQ.err; echo "sq n128 rc=$?"
cd $R
python scripts/prof_summary.py r03_final 30000001 | tail -40
PROF_NAME=n128 PROF_STATES=128 PROF_CMD="python scripts/r03/n128_run.py (config 5: 30 M bins x 128 states; full-count E-steps, then factored ones)" python scripts/prof_summary.py r03_final 30000001 | tail -40
python - <<'PY'
import csv, collections, json, re
rows = list(csv.DictReader(open('gpurun_out/pmc/n128_SQ_counter_collection.csv')))
d = collections.defaultdict(dict); dur = {}
for r in rows:
    nm = r['Kernel_Name']; m = re.search(r'(k_[a-z0-9_]+)', nm); k = m.group(1) if m else nm[:20]
    d[(r['Dispatch_Id'], k)][r['Counter_Name']] = float(r['Counter_Value'])
    dur[(r['Dispatch_Id'], k)] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
best = {}
for key, c in d.items():
    k = key[1]
    if k not in best or dur[key] > best[k][1]: best[k] = (c, dur[key])
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -- python scripts/r03/n128_run.py 2",
       "note": "the longest launch of every kernel (config 5: 30 M bins x 128 states); the SQ counters cover ~0.8 of the device (see profiles/r02_sq_counters.json): ratios are what they are good for",
       "kernels": {}}
for k, (c, t) in sorted(best.items(), key=lambda x: -x[1][1]):
    if t > 0.05:
        e = dict(ms=round(t, 3), **{a: b for a, b in sorted(c.items())})
        if c.get("SQ_INSTS_MFMA"): e["busy_cycles_per_mfma"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["SQ_INSTS_MFMA"]; e["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"]
        out["kernels"][k] = e
        print('%-24s %8.3f ms ' % (k, t), {a: '%.3g' % b for a, b in e.items() if a != "ms"})
json.dump(out, open('profiles/r03_n128_sq_counters.json', 'w'), indent=1)
PY
find gpurun_out/pmc gpurun_out/prof -name "*.csv" -size +3M -delete
