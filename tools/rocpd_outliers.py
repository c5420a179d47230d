#!/usr/bin/env python
"""usage: python tools/rocpd_outliers.py <results.db> [n] -- the n longest kernel dispatches of a rocprofv3 --kernel-trace
database, and the largest gaps between consecutive dispatches (where a stalled step shows up)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = list(db.execute("select name, start, end from kernels order by start"))
print("longest dispatches (us):")
for name, s, e in sorted(rows, key=lambda r: r[1] - r[2])[:n]:
    print(f"  {(e - s) / 1e3:10.1f}  {name[:70]}")
gaps = sorted(((rows[i + 1][1] - rows[i][2], rows[i][0], rows[i + 1][0]) for i in range(len(rows) - 1)), reverse=True)[:n]
print("largest gaps between consecutive dispatches (us):")
for g, a, b in gaps:
    print(f"  {g / 1e3:10.1f}  after {a[:40]} before {b[:40]}")
