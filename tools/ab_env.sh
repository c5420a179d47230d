#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh VAR "v1 v2 ..." [bench args] -- bench.py once per value of the environment
# variable (plus once with it unset, first and last); prints iters/s and the per-view stage times
var=$1; vals=$2; shift 2
run() { python bench.py --steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$var=${!var}', d['value'], d.get('stage_ms_per_view'))"; }
unset $var; run "$@"
for v in $vals; do export $var=$v; run "$@"; done
unset $var; run "$@"
