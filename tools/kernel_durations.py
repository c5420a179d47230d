#!/usr/bin/env python
"""Per-dispatch durations (us, in launch order) of one kernel from a rocprofv3 rocpd database.
usage: tools/kernel_durations.py <results.db> <kernel-name-substring>"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tables if "kernels" == t]
src = "kernels" if "kernels" in tables else kd[0]
cols = [r[1] for r in db.execute(f"pragma table_info({src})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(db.execute(f"select {name_col}, start, end from {src} where {name_col} like ? order by start", (f"%{sys.argv[2]}%",)))
print(len(rows), "dispatches:", " ".join("%.0f" % ((e - s) / 1e3) for _n, s, e in rows))
