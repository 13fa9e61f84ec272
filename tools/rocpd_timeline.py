#!/usr/bin/env python3
"""Launch gaps from a rocprofv3 --kernel-trace rocpd database: for every pair of consecutive dispatches (previous kernel ->
next kernel), the median gap between the end of one and the start of the next, the count, and the kernels' median durations.
usage: python tools/rocpd_timeline.py <results.db> [min_count]"""
import sqlite3, sys, statistics as st
db = sys.argv[1]; min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
s_col = "start" if "start" in cols else "start_timestamp"; e_col = "end" if "end" in cols else "end_timestamp"
rows = cur.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("obb::", "")[:60]
pairs, durs = {}, {}
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    pairs.setdefault((short(n0), short(n1)), []).append((s1 - e0) / 1e3)
for n, s, e in rows:
    durs.setdefault(short(n), []).append((e - s) / 1e3)
print("| previous kernel | next kernel | pairs | median gap us | p90 gap us | median duration of next us |")
print("|---|---|---:|---:|---:|---:|")
for (a, b), g in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
    if len(g) < min_count: continue
    g.sort()
    print(f"| `{a}` | `{b}` | {len(g)} | {st.median(g):.2f} | {g[int(len(g) * 0.9)]:.2f} | {st.median(durs[b]):.2f} |")
