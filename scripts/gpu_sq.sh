#!/bin/bash
# SQ counters of the E-step kernels (one pass, --kernel-trace + --pmc only): instructions, busy and wait cycles per launch
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp
CNT="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM"
timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/sq -o full -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --exact-extra 0 > $R/gpurun_out/sq/full.json 2> $R/gpurun_out/sq/full.err
echo "full exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/sq -o fact -- python $R/scripts/sweep_factored.py "" > $R/gpurun_out/sq/fact.log 2> $R/gpurun_out/sq/fact.err
echo "fact exit $?"
cd $R
python scripts/pmc_summary.py gpurun_out/sq > gpurun_out/sq/summary.txt
python - <<'PY'
import csv, glob, collections, re
for f in sorted(glob.glob('gpurun_out/sq/**/*kernel_trace.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']; m = re.search(r'(k_[a-z0-9_]+)', nm)
        k = m.group(1) if m else nm[:30]
        agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    print('##', f)
    for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
        print('%-28s launches=%4d avg_ms=%9.3f total_ms=%9.2f' % (k, n, t / n, t))
PY
find gpurun_out/sq -name "*.csv" -size +2M -delete
head -150 gpurun_out/sq/summary.txt
