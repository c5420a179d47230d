"""Builds libb3gs_raster.so in-tree with hipcc for gfx950 (no torch extension machinery, no hipify).
`python -m binocular3dgs_amd.build` or __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libb3gs_raster.so")


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libb3gs_raster.so failed (see log above)")
    if not os.path.exists(LIB):
        raise RuntimeError(f"{LIB} was not produced")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
