#!/usr/bin/env python
"""Busy fraction / overlap statistics of a rocprofv3 kernel trace (rocpd sqlite).
usage: tools/timeline.py <results.db> [skip_fraction]"""
import re
import sqlite3
import sys
import numpy as np

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = list(cur.execute("select name, start, end from kernels order by start"))
names = [re.sub(r"\(anonymous namespace\)::", "", r[0]).split("(")[0].split("<")[0][-28:] for r in rows]
st = np.array([r[1] for r in rows], dtype=np.int64)
en = np.array([r[2] for r in rows], dtype=np.int64)
# keep the steady state: last 60 % of the trace
t0 = st.min() + int(0.4 * (en.max() - st.min()))
keep = st >= t0
st, en, names = st[keep], en[keep], [n for n, k in zip(names, keep) if k]
span = en.max() - st.min()
# union of intervals
order = np.argsort(st)
busy, cur_s, cur_e = 0, st[order[0]], en[order[0]]
for i in order[1:]:
    if st[i] > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = st[i], en[i]
    else:
        cur_e = max(cur_e, en[i])
busy += cur_e - cur_s
print(f"span {span/1e3:.1f} us  busy(union) {busy/1e3:.1f} us = {busy/span:.3f}  sum of kernel durations {(en-st).sum()/1e3:.1f} us  "
      f"avg concurrency {(en-st).sum()/busy:.2f}")
agg = {}
for n, s, e in zip(names, st, en):
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n:30s} calls {c:5d} total {t/1e3:9.1f} us avg {t/c/1e3:7.1f} us  share of busy {t/busy:.2f}")
