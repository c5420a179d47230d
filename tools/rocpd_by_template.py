#!/usr/bin/env python
"""usage: python tools/rocpd_by_template.py <results.db> <name-substring> -- calls / average duration (us) of every kernel
whose name contains the substring, one line per FULL name (template arguments kept: rocpd_summary.py folds them)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if sys.argv[2] in name:
        print(f"{calls:6d} calls  avg {avg / 1e3:8.2f} us  {name[:150]}")
