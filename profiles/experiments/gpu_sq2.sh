#!/bin/bash
# SQ counters + isolated durations for one option set of profiles/experiments/sweep_factored.py (ARGS), counters in CNT
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq2; rm -f $R/gpurun_out/sq2/*
cd /tmp
CNT=${CNT:-"SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"}
timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/sq2 -o run -- python $R/profiles/experiments/sweep_factored.py ${ARGS:---full fuse=1} > $R/gpurun_out/sq2/run.log 2> $R/gpurun_out/sq2/run.err
echo "exit $?"; tail -3 $R/gpurun_out/sq2/run.err | cut -c1-300
cd $R
python - <<'PY'
import csv, collections, re
rows=list(csv.DictReader(open('gpurun_out/sq2/run_counter_collection.csv')))
d=collections.defaultdict(dict); dur={}
for r in rows:
    nm=r['Kernel_Name']; m=re.search(r'(k_[a-z0-9_]+)',nm); k=(m.group(1) if m else nm[:20])+('<rep>' if 'Lb1' in nm else '')
    d[(r['Dispatch_Id'],k)][r['Counter_Name']]=float(r['Counter_Value'])
    dur[(r['Dispatch_Id'],k)]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
best={}
for key,c in d.items():
    k=key[1]
    if k not in best or dur[key]>best[k][1]: best[k]=(c,dur[key])
for k,(c,t) in sorted(best.items(), key=lambda x:-x[1][1]):
    if t>0.05: print('%-22s %8.3f ms '%(k,t), {a:'%.3g'%b for a,b in sorted(c.items())})
PY
find gpurun_out/sq2 -name "*.csv" -size +2M -delete
