#!/usr/bin/env python3
"""Where the host blocks: from a rocprofv3 --hip-trace --kernel-trace rocpd database (markdown).
usage: python tools/rocpd_hiptrace.py <results.db> [long_call_us=300] [idle_gap_us=300] [last_ms=0]
 (last_ms > 0: sections 2 and 3 only look at the last `last_ms` milliseconds of the trace -- the timed loops, not the warm-up)
 1. HIP API totals (count, total, mean, max);
 2. every API call longer than long_call_us: when, how long, and the kernels that ran meanwhile;
 3. every GPU idle gap longer than idle_gap_us between two dispatches: the neighbours and the API calls the host made meanwhile."""
import sqlite3
import sys

db = sys.argv[1]
long_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
last_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
cur = sqlite3.connect(db).cursor()
short = lambda n: n.split("(")[0].replace("void ", "")[:70]
regs = cur.execute("select name, category, start, end, tid from regions order by start").fetchall()
kerns = cur.execute("select name, start, end from kernels order by start").fetchall()
if not regs:
    print("no regions in the database (was --hip-trace given?)"); sys.exit(0)
t0 = min(regs[0][2], kerns[0][1] if kerns else regs[0][2])
t_end = max(max(r[3] for r in regs), kerns[-1][2] if kerns else 0)
t_from = (t_end - last_ms * 1e6) if last_ms > 0 else (t0 + 1e9)
print(f"# HIP API timeline summary of {db}\n")
print(f"{len(regs)} API regions, {len(kerns)} kernel dispatches, span {(max(r[3] for r in regs) - t0) / 1e6:.1f} ms\n")
tot = {}
for n, c, s, e, tid in regs:
    d = tot.setdefault(n, [0, 0, 0])
    d[0] += 1; d[1] += e - s; d[2] = max(d[2], e - s)
print("## API totals\n\n| call | count | total ms | mean us | max us |\n|---|---:|---:|---:|---:|")
for n, (k, t, m) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"| {n} | {k} | {t / 1e6:.2f} | {t / k / 1e3:.1f} | {m / 1e3:.1f} |")
print(f"\n## API calls longer than {long_us:.0f} us (first 60 of the window looked at)\n")
print("| at ms | call | us | kernels running meanwhile |\n|---:|---|---:|---|")
shown = 0
for n, c, s, e, tid in regs:
    if (e - s) / 1e3 < long_us or s < t_from:
        continue
    inside = {}
    for kn, ks, ke in kerns:
        if ke > s and ks < e:
            inside[short(kn)] = inside.get(short(kn), 0) + 1
    desc = ", ".join(f"{k} x{v}" for k, v in sorted(inside.items(), key=lambda kv: -kv[1])[:4])
    print(f"| {(s - t0) / 1e6:.2f} | {n} | {(e - s) / 1e3:.0f} | {desc} |")
    shown += 1
    if shown >= 60:
        break
print(f"\n## GPU idle gaps longer than {gap_us:.0f} us (first 60 of the window looked at)\n")
print("| at ms | gap us | previous kernel | next kernel | host API calls in the gap (longest first) |\n|---:|---:|---|---|---|")
shown = 0
ri = 0
for (n0, s0, e0), (n1, s1, e1) in zip(kerns, kerns[1:]):
    if (s1 - e0) / 1e3 < gap_us or e0 < t_from:
        continue
    calls = [(e - s, n) for n, c, s, e, tid in regs if e > e0 and s < s1]
    calls.sort(reverse=True)
    desc = ", ".join(f"{n} {d / 1e3:.0f}us" for d, n in calls[:5])
    print(f"| {(e0 - t0) / 1e6:.2f} | {(s1 - e0) / 1e3:.0f} | {short(n0)} | {short(n1)} | {desc} |")
    shown += 1
    if shown >= 60:
        break
