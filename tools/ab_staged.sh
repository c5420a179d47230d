#!/bin/bash
# usage (GPU box, repo root): tools/ab_staged.sh -- staged marks (GeomView::staged) on / off: parity subset with the marks on, then
# four alternating rounds of the headline (value ms_per_step, per-view stage times)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bwd_kernels.py tests/test_gpu_fused.py tests/test_gpu_bench_config.py tests/test_gpu_dropin_pair.py -x -q 2>&1 | tail -1
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('NO_STAGED=${B3GS_NO_STAGED:-unset}', d['value'], d['ms_per_step'], d['stage_ms_per_view'])"; }
for rep in 1 2 3 4; do export B3GS_NO_STAGED=1; run; unset B3GS_NO_STAGED; run; done
