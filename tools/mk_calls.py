#!/usr/bin/env python3
"""Dispatches of the last nms_rotated call in a rocprofv3 --kernel-trace rocpd database, in launch order: start offset,
duration and the gap to the previous kernel.   python tools/mk_calls.py <results.db> [first kernel name fragment]"""
import sqlite3, sys
db = sys.argv[1]; first = sys.argv[2] if len(sys.argv) > 2 else "k_ps_local_scores"
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
s_col = "start" if "start" in cols else "start_timestamp"; e_col = "end" if "end" in cols else "end_timestamp"
rows = cur.execute(f"select name, {s_col}, {e_col}, grid_x, workgroup_x from kernels order by {s_col}").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("obb::", "")[:48]
idx = [i for i, r in enumerate(rows) if first in r[0]]
if not idx:
    sys.exit("no such kernel")
rows = rows[idx[-1]:]
t0 = rows[0][1]; prev_end = None; tot = 0.0
print(f"{'kernel':50s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}  grid")
for n, s, e, gx, wx in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short(n):50s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  {gx // max(1, wx)} x {wx}")
    prev_end = e; tot += (e - s) / 1e3
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, kernels {tot:.1f} us, {len(rows)} dispatches")
