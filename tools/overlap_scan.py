#!/usr/bin/env python
"""Line-overlap scan of every source in this repo against every source of the reference (build container only: needs
/root/reference).  A line counts when, with all whitespace removed, it has >= 12 characters, is not a pure comment / import,
and appears in the reference file.  Prints repo files with the most shared lines per reference file.
    python tools/overlap_scan.py [min_shared]"""
import os
import re
import sys

REF, REPO = "/root/reference", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".py", ".hip", ".h", ".c", ".cpp", ".cu", ".cuh", ".sh")


def norm_lines(path):
    out = set()
    try:
        text = open(path, errors="ignore").read()
    except OSError:
        return out
    for ln in text.splitlines():
        s = re.sub(r"\s+", "", ln)
        if len(s) < 12 or s.startswith(("#", "//", "import", "from", '"""', "*")):
            continue
        out.add(s)
    return out


def walk(root, skip=()):
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in (".git", "__pycache__", "gpurun_out", "dense_matcher") and x not in skip]
        for f in files:
            if f.endswith(EXT):
                yield os.path.join(d, f)


def main():
    thr = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ref = {p: norm_lines(p) for p in walk(REF)}
    rows = []
    for p in walk(REPO):
        mine = norm_lines(p)
        if not mine:
            continue
        for rp, rl in ref.items():
            n = len(mine & rl)
            if n >= thr:
                rows.append((n, os.path.relpath(p, REPO), os.path.relpath(rp, REF), round(n / max(len(mine), 1), 2)))
    for r in sorted(rows, reverse=True):
        print(*r)
    if not rows:
        print(f"no repo file shares >= {thr} lines with any reference file")


if __name__ == "__main__":
    main()
