#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel average counter value per launch."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
def short(nm):
    if 'k_fwd_fast' in nm: return 'k_fwd_fast<%s>' % ('repair' if 'Lb1' in nm else 'speculate')
    if 'k_bwd_fast' in nm: return 'k_bwd_fast<%s>' % ('repair' if 'Lb1' in nm else 'speculate')
    if 'k_verify' in nm: return 'k_verify'
    m = re.search(r'(k_[a-z0-9_]+?)E', nm)
    return m.group(1) if m and 'psmc' in nm else nm[:40]
for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    for r in rows:
        k = (short(r.get('Kernel_Name', '')), r.get('Counter_Name', ''))
        agg[k][0] += 1; agg[k][1] += float(r.get('Counter_Value', 0))
    print('##', os.path.relpath(f, d))
    for (k, c), (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print('%-28s %-12s launches=%4d  avg_per_launch=%16.1f  total=%18.1f' % (k, c, n, v / n, v))
