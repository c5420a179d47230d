#!/bin/bash
# usage: tools/kernel_resources.sh [file.hip ...] -- per-kernel VGPR / SGPR / scratch / LDS of the gfx950 code objects
# (cross-compiles to assembly; run it after every kernel refactor: a by-value argument block that the compiler copies
# to scratch memory shows up as private_segment_fixed_size > 0)
cd "$(dirname "$0")/../binocular3dgs_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -mllvm -amdgpu-atomic-optimizer-strategy=None $EXTRA"
for f in ${@:-api.hip preprocess.hip binning.hip render.hip optim.hip loss.hip densify.hip knn.hip}; do
  /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only $f -o /tmp/kr_$$.s 2>/dev/null
  python3 - /tmp/kr_$$.s $f <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    lds, name, body = m.group(1), m.group(2), m.group(3)
    g = lambda k: (re.search(k + r':\s+(\d+)', body) or [None, '?'])[1]
    short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:64]
    print(f"{sys.argv[2]:16s} {short:64s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {lds:>6s}")
PY
done
rm -f /tmp/kr_$$.s
