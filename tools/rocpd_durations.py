#!/usr/bin/env python
"""usage: python tools/rocpd_durations.py <results.db> <name-substring> -- every dispatch's duration (us) of the kernels whose
name contains the substring, in launch order (rocprofv3 --kernel-trace rocpd database)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
cand = [v for v in views if v == "kernels"] or [v for v in views if "kernel" in v.lower()]
for v in cand:
    cols = [c[1] for c in db.execute(f"pragma table_info({v})")]
    if "name" in cols and "start" in cols and "end" in cols:
        rows = list(db.execute(f"select name, start, end from {v} where name like ? order by start", (f"%{sys.argv[2]}%",)))
        d = [(e - s) / 1e3 for _, s, e in rows]
        print(v, len(d), "dispatches; us:", " ".join(f"{x:.1f}" for x in d))
        break
else:
    print("no kernel view with name/start/end; views:", views)
    for v in cand:
        print(v, [c[1] for c in db.execute(f"pragma table_info({v})")])
