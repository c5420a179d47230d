#!/bin/bash
# usage (GPU box, repo root): tools/ab_lib.sh -- alternating headline runs: tools/ab/base.so (B3GS_LIB) against the in-tree library
run() { B3GS_LIB=$1 python bench.py --steps 40 --warmup 5 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${2}', d['value'], d['ms_per_step'], d['stage_ms_per_view']['preprocess_bwd'])"; }
for rep in 1 2 3 4; do run $PWD/tools/ab/${AB_NAME:-base}.so ${AB_NAME:-base}; run "" new; done
