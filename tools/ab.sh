#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh [bench args] -- runs bench.py once per tools/ab/*.so (B3GS_LIB) and once with the
# in-tree library; prints iters/s and the per-view stage times
run() { B3GS_LIB=$1 python bench.py --steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['stage_ms_per_view'])"; }
run "" "$@"
for f in tools/ab/*.so; do run $PWD/$f "$@"; done
run "" "$@"
