#!/bin/bash
# usage: tools/stage_times.sh [bench args]  -> one compact line: iters/s, ms/step, per-stage ms per view
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'it/s', d['ms_per_step'], 'ms/step', d['stage_ms_per_view'], 'N', d['config']['instances_N'], 'Nbinned', d['config'].get('instances_N_binned'))"
