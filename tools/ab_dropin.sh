#!/bin/bash
# usage (GPU box, repo root): tools/ab_dropin.sh -- the reference-shaped surface (bench.py --path dropin) once per tools/ab/*.so
# and twice with the in-tree library: iters/s and per-view stage times
run() { B3GS_LIB=$1 python bench.py --path dropin --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['stage_ms_per_view'])"; }
run "" "$@"
for f in tools/ab/*.so; do run $PWD/$f "$@"; done
run "" "$@"
