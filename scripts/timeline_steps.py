#!/usr/bin/env python3
"""One line per E-step of a rocprofv3 --kernel-trace run (rocpd database): when each kernel of the step started / ended, relative to
the step's first kernel.  Complements scripts/prof_timeline.py (which lists the last step kernel by kernel): bimodal steps show here.

    python scripts/timeline_steps.py gpurun_out/prof/tl_results.db [k_reduce2|k_reduce_factored2] [first_step]
"""
import sqlite3, sys, re, glob
db = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*.db"))[-1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_reduce2"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (disp, sym)))
def short(nm):
    m = re.search(r'psmc::(k_[a-z0-9_]+)', nm) or re.search(r'_ZN4psmc\d+(k_[a-z0-9_]+?)I', nm) or re.search(r'(k_[a-z0-9_]+)', nm)
    return m.group(1) if m else nm[:20]
steps, cur_step = [], []
for r in rows:
    cur_step.append(r)
    if key in r[2]: steps.append(cur_step); cur_step = []
names = ["k_walk1_struct", "k_sweep_struct", "k_fwd_struct", "k_bwd_struct", "k_kcol2_struct", "k_kchain_struct", "k_bwd_count4f_struct", "k_bwd_acc_ckpt", "k_verify", "k_reduce"]
print("step  total | " + " | ".join("%-13s" % n[2:15] for n in names))
for i, st in enumerate(steps):
    if i < first: continue
    t0 = min(r[0] for r in st if "fillBuffer" not in r[2]) if any("fillBuffer" not in r[2] for r in st) else st[0][0]
    cells = []
    for n in names:
        rr = [r for r in st if short(r[2]).startswith(n)]
        cells.append("%5.2f-%5.2f %d" % ((min(r[0] for r in rr) - t0) / 1e6, (max(r[1] for r in rr) - t0) / 1e6, len(rr)) if rr else " " * 13)
    print("%3d  %6.3f | %s" % (i, (max(r[1] for r in st) - t0) / 1e6, " | ".join("%-13s" % c for c in cells)))
