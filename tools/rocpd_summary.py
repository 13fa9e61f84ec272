#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace [--stats] [--pmc ...]) as a markdown table.

usage: python tools/rocpd_summary.py <results.db> [title] > profiles/<name>.md
(rocprofv3 in this image writes an SQLite `rocpd` database by default; this prints the same per-kernel
statistics as its `--stats` CSV: calls, total / average / min / max duration, share of GPU kernel time,
plus registers / LDS of each kernel and, when the run collected counters, the per-kernel counter sums.)
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    cur = c.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x), min(grid_x), max(grid_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | wg | grid.x (threads) |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    for r in rows:
        name = r[0]
        if len(name) > 110:
            name = name[:107] + "..."
        gx = f"{r[11]}" if r[11] == r[12] else f"{r[11]}..{r[12]}"
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
              f"{100.0 * r[2] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {gx} |")
    print(f"\ntotal GPU kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    try:
        pm = cur.execute("select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p join kernels k "
                         "on p.event_id = k.id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if not pm:
        try:
            pm = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                             "group by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            pm = []
    if pm:
        print("\n## counters (sum over dispatches; per-dispatch = sum / dispatches)\n")
        print("| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---:|---:|---:|")
        for k, cn, n, v in pm:
            if len(k) > 90:
                k = k[:87] + "..."
            print(f"| `{k}` | {cn} | {n} | {v:.6g} | {v / max(1, n):.6g} |")


if __name__ == "__main__":
    main()
