#!/bin/bash
# isolated duration of the bulk forward sweep (rocprofv3 --pmc serialises the kernels) for library variants
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "" NO_IO NO_CKG NO_EV; do
  mkdir -p /tmp/ab$V; cd /tmp
  LIB=$R/psmc_amd/libpsmc_hip.so; [ -n "$V" ] && LIB=$R/psmc_amd/libpsmc_hip_$V.so
  PSMC_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d /tmp/ab$V -o run -- python $R/scripts/sweep_factored.py --full "" > /tmp/ab$V/log 2>&1
  python - "$V" <<'PY'
import csv, sys, collections, re
v = sys.argv[1]
d = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/ab%s/run_kernel_trace.csv' % v)):
    m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name']); k = (m.group(1) if m else '?') + ('<rep>' if 'Lb1' in r['Kernel_Name'] else '')
    d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print("%-8s" % (v or "base"), " ".join("%s %.2f" % (k, max(x)) for k, x in d.items() if k in ('k_fwd_struct', 'k_bwd_struct', 'k_bwd_count4f_struct', 'k_kcol_struct')))
PY
done
