#!/bin/bash
# usage (GPU box, repo root): tools/ab_flush.sh VAR "v1 v2 .." -- parity of each value of a blend-backward A/B switch (bwd kernel, fused and
# exact-bench-configuration tests), then three alternating rounds of the headline: value ms_per_step render_bwd avg_launch_ms
var=$1; vals=$2
for f in $vals; do
  export $var=$f
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bwd_kernels.py tests/test_gpu_fused.py tests/test_gpu_bench_config.py -x -q 2>&1 | tail -1
done
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$var=${!var}', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do for f in $vals; do export $var=$f; run; done; done
