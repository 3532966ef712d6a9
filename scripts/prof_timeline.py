#!/usr/bin/env python3
"""Kernel timeline of the LAST E-step in a rocprofv3 --kernel-trace run of bench.py (rocpd database)."""
import sqlite3, sys, re, glob, os
db = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*.db"))[-1]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = "d.start, d.end, s.kernel_name" + (", d.%s" % qcol if qcol else ", 0") + (", d.%s" % scol if scol else ", 0") + ", d.grid_size_x" if "grid_size_x" in cols else "d.start, d.end, s.kernel_name, 0, 0, 0"
rows = list(cur.execute("select %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (sel, disp, sym)))
def short(nm):
    m = re.search(r'psmc::(k_[a-z0-9_]+)(<[^>]*>)?', nm) or re.search(r'_ZN4psmc\d+(k_[a-z0-9_]+?)I?L?b?(\d?)E', nm)
    return (m.group(1) + (m.group(2) or "")) if m else nm.split('(')[0][:30]
key = sys.argv[2] if len(sys.argv) > 2 else "k_reduce2"
ends = [i for i, r in enumerate(rows) if key in r[2]]
if len(ends) < 2: print("not enough E-steps"); sys.exit(0)
lo, hi = ends[-2] + 1, ends[-1]
t0 = min(r[0] for r in rows[lo:hi + 1])
print("%-34s %9s %9s %9s %5s %5s %8s" % ("kernel", "start_ms", "end_ms", "dur_ms", "queue", "strm", "grid"))
for r in rows[lo:hi + 1]:
    if (r[1] - r[0]) < 20000 and "struct" not in r[2] and "expect" not in r[2] and "all" not in sys.argv[3:]: continue
    print("%-34s %9.3f %9.3f %9.3f %5s %5s %8s" % (short(r[2])[:34], (r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[3], r[4], r[5]))
