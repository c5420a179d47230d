#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the compact per-kernel CSV kept under profiles/.
usage: python tools/rocpd_summary.py gpurun_out/prof1/*/*_results.db profiles/r01_xxx_kernel_stats.csv"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    base = m.group(1) if m else name
    if "multi_tensor_apply_kernel" in name and "FusedAdam" in name:
        base = "torch::fused_adam"
    elif base.startswith("at::native::") and "<" in name:
        inner = re.findall(r"(\w+_kernel_cuda|\w+Functor\w*|CatArrayBatchedCopy\w*|reduce_kernel|masked_fill\w*|where_kernel\w*|launch_clamp\w*|compare_scalar\w*)", name)
        base = "torch::" + (inner[0] if inner else base.split("::")[-1])
    return base[:60]


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += total
        a[2] += pct
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, f"{total:.1f}", f"{total / calls:.2f}", f"{pct:.2f}"])
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
