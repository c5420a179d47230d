#!/usr/bin/env python
"""usage: tools/isa_count.py [kernel-substring ...] -- static instruction mix of the loops of the blend kernels (gfx950
assembly from hipcc -S): per loop body the VALU / transcendental / DPP / LDS / SALU / VMEM / s_waitcnt counts.  The hot
loop of render_bwd_kernel<64,false> is unrolled by two (two candidates per trip).  Cross-check with the counters:
    SQ_INSTS_VALU per launch ~ (candidates evaluated per launch, tools/bwd_trace_batched.py) x (VALU per candidate) + staging
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-fast-math "
         "-mllvm -amdgpu-atomic-optimizer-strategy=None").split()


def classify(seg):
    c = Counter()
    for x in seg:
        x = x.strip()
        if not x or x.startswith((".", ";")) or x.endswith(":"):
            continue
        op = x.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if any(k in op for k in ("exp", "rcp", "log", "sqrt", "rsq")):
                c["valu_transcendental"] += 1
            if "dpp" in x:
                c["valu_dpp"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
            c["lds:" + op] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            c["vmem"] += 1
            c["vmem:" + op] += 1
    return c


def main():
    wanted = sys.argv[1:] or ["render_bwd_kernelILb0ELi64ELb1ELb1", "render_fwd_kernelILi256ELb0"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "render.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", os.path.join(ROOT, "binocular3dgs_amd/csrc/render.hip"),
                        "-o", asm], check=True, stderr=subprocess.DEVNULL)
        L = open(asm).read().split("\n")
    for w in wanted:
        start = next(i for i, l in enumerate(L) if re.match(r"^_ZN\S*" + re.escape(w) + r"\S*:", l))
        end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
        lines = L[start:end]
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        print(f"== {L[start].split(':')[0]}  ({len(lines)} lines; whole kernel: {dict(classify(lines))})")
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i:
                    loops.append((labels[t], i, t))
        for a, b, t in sorted(set(loops), key=lambda x: x[1] - x[0]):
            print(f"  loop {t} lines {a}-{b}: {dict(classify(lines[a:b + 1]))}")


if __name__ == "__main__":
    main()
