"""Development aid: longest dispatches / largest gaps in a rocprofv3 kernel trace (rocpd database)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); cur = c.cursor()
rows = cur.execute("select name, start, end, duration from kernels order by start").fetchall()
print("dispatches", len(rows))
top = sorted(rows, key=lambda r: -r[3])[:8]
for r in top: print(f"{r[3]/1e3:10.1f} us  {r[0][:90]}")
gaps = sorted(((rows[i+1][1]-rows[i][2], rows[i][0][:50], rows[i+1][0][:50]) for i in range(len(rows)-1)), reverse=True)[:8]
for g in gaps: print(f"gap {g[0]/1e3:10.1f} us after {g[1]} before {g[2]}")
