"""Builds, in-tree (csrc/Makefile): libb3gs_raster.so with hipcc for gfx950 (no torch extension machinery, no hipify) and
the compiled python module `_C` (csrc/host/*.cpp: host-only C++, g++ against the torch headers, links libb3gs_raster.so).
`python -m binocular3dgs_amd.build` or __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libb3gs_raster.so")


def ext_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), f"PYTHON={sys.executable}"]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("build of libb3gs_raster.so / the _C module failed (see log above)")
    for out in (LIB, ext_path()):
        if not os.path.exists(out):
            raise RuntimeError(f"{out} was not produced")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
