#!/bin/bash
# usage (GPU box): tools/sizes_ab.sh -- one round vs two rounds (B3GS_SEG1_FRAC) at the Gaussian counts of BASELINE's configs
for P in 100000 250000 500000 1000000 2000000; do
for F in 0 0.125; do
B3GS_SEG1_FRAC=$F python bench.py --gaussians $P --no-extras --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']; print('P=$P frac=$F:', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'N_binned/view', c.get('instances_N_binned'), 'N_ref', c.get('instances_N'))"
done; done
