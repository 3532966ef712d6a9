#!/bin/bash
# round 2, GPU call T: final evidence with the final build -- whole GPU suite, EM parity over 25 rounds, the bench line
# as the driver runs it, rocprofv3 kernel stats / SQ counters / PMC traffic of the bench command, config-4 timing.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/t_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/t_pytest.log | cut -c1-250
timeout 600 python scripts/em_parity.py gpurun_out/t_em_parity.json gpurun_out/t_traj_n64.json > gpurun_out/t_em_parity.out 2> gpurun_out/t_em_parity.err
echo "em_parity rc=$?"; tail -7 gpurun_out/t_em_parity.err | cut -c1-250
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/t_bench.json 2> gpurun_out/t_bench.err
echo "bench rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/t_bench.json')); print(r['ms_per_step'], r['value'], r['steady_state']['ms_per_step'], r['roofline']['frac'], r['roofline']['kernel'], r['factored_stats']['ms_per_step'], r['exact_mode']['ms_per_step'], r['n128']['ms_per_step'], r['n128']['factored_stats']['ms_per_step'])"
rm -rf gpurun_out/prof gpurun_out/pmc
timeout 300 bash scripts/gpu_prof.sh > gpurun_out/t_prof.log 2>&1
echo "prof rc=$?"
mkdir -p gpurun_out/sqb; rm -f gpurun_out/sqb/*
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/sqb -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 0 > $R/gpurun_out/sqb/bench.json 2> $R/gpurun_out/sqb/bench.err )
echo "sq rc=$?"
python - <<'PY'
import csv, collections, re, json, glob
fs = glob.glob('gpurun_out/sqb/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0]))) if fs else []
d = collections.defaultdict(dict); dur = {}
for r in rows:
    nm = r['Kernel_Name']; m = re.search(r'(k_[a-z0-9_]+)', nm); k = (m.group(1) if m else nm[:24])
    key = (r['Dispatch_Id'], k)
    d[key][r['Counter_Name']] = float(r['Counter_Value'])
    dur[key] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); tms = collections.Counter(); big = collections.Counter()
for key, c in d.items():
    cnt[key[1]] += 1; tms[key[1]] += dur[key]; big[key[1]] += dur[key] > 1.0
    for a, b in c.items(): agg[key[1]][a] += b
out = {k: dict(launches=cnt[k], launches_over_1ms=big[k], total_ms=round(tms[k], 3), **{a: b for a, b in sorted(agg[k].items())}) for k in sorted(agg, key=lambda k: -tms[k]) if tms[k] > 0.05}
json.dump(out, open('gpurun_out/t_sq_counters_raw.json', 'w'), indent=1)
PY
find gpurun_out/sqb -name "*.csv" -size +4M -delete
timeout 500 bash scripts/gpu_pmc.sh > gpurun_out/t_pmc.log 2>&1
echo "pmc rc=$?"

BOOT_FAST_ONLY=1 timeout 400 python scripts/time_boot.py gpurun_out/t_boot_timing.json > gpurun_out/t_boot.out 2>&1
echo "boot rc=$?"; tail -3 gpurun_out/t_boot.out | cut -c1-300
