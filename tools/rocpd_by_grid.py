#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 (rocpd sqlite) kernel trace: average duration grouped by (kernel, grid size).
Tells the depth-sort passes (3 x 245 workgroups) from the tile-split passes (6 x ~1200) of the same radix kernels.
usage: python tools/rocpd_by_grid.py <results.db> [name substring ...]"""
import sqlite3
import sys


def main(db_path, pats):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tabs if "kernel_dispatch" in t]
    if not disp:
        print("tables:", tabs)
        return
    t = disp[0]
    cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
    sym = [x for x in tabs if "kernel_symbol" in x or "kernel_symbols" in x]
    print("dispatch table", t, cols)
    if sym:
        print("symbol table", sym[0], [r[1] for r in db.execute(f"pragma table_info({sym[0]})")])
    namecol = "kernel_name" if "kernel_name" in [r[1] for r in db.execute(f"pragma table_info({sym[0]})")] else "display_name"
    q = (f"select s.{namecol}, d.grid_size_x, d.grid_size_y, count(*), avg(d.end - d.start) / 1000.0 "
         f"from {t} d join {sym[0]} s on d.kernel_id = s.id group by 1, 2, 3 order by 1, 2")
    for name, gx, gy, n, avg in db.execute(q):
        if pats and not any(p in name for p in pats):
            continue
        print(f"{name[:70]:70s} grid=({gx},{gy}) calls={n:5d} avg_us={avg:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
