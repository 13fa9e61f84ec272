#!/usr/bin/env python3
"""Per-kernel counter sums from a rocprofv3 rocpd database collected with --pmc (markdown)."""
import sqlite3, sys
db, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
c = sqlite3.connect(db); cur = c.cursor()
cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
print(f"# {title}\n")
print("<!-- pmc_events columns:", cols, "-->")
name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
cnt_col = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
val_col = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
if not (cnt_col and val_col):
    print("unexpected schema"); sys.exit(0)
if name_col:
    q = f"select {name_col}, {cnt_col}, count(*), sum({val_col}), min({val_col}), max({val_col}) from pmc_events group by {name_col}, {cnt_col} order by sum({val_col}) desc"
else:
    q = (f"select k.name, p.{cnt_col}, count(*), sum(p.{val_col}), min(p.{val_col}), max(p.{val_col}) from pmc_events p join kernels k on p.event_id = k.event_id "
         f"group by k.name, p.{cnt_col} order by sum(p.{val_col}) desc")
rows = cur.execute(q).fetchall()
print("| kernel | counter | dispatches | sum | per dispatch | min | max |")
print("|---|---|---:|---:|---:|---:|---:|")
for k, cn, n, v, mn, mx in rows:
    k = k if len(k) <= 100 else k[:97] + "..."
    if v is None: continue
    print(f"| `{k}` | {cn} | {n} | {v:.6g} | {v / max(1, n):.6g} | {mn:.6g} | {mx:.6g} |")

# per-dispatch values (launch order) of this project's kernels: lets one separate workloads that share a kernel
print("\n## per-dispatch values of obb:: kernels (launch order)\n")
if name_col:
    q = f"select {name_col}, {cnt_col}, {val_col} from pmc_events where {name_col} like '%obb::%' order by start"
    per = {}
    for k, cn, v in cur.execute(q):
        per.setdefault((k, cn), []).append(v)
    for (k, cn), vs in per.items():
        k = k if len(k) <= 80 else k[:77] + "..."
        print(f"- `{k}` {cn}: " + ", ".join(f"{v:.6g}" for v in vs[:24]))
