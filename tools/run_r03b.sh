#!/bin/bash
python -m pytest tests/test_gpu_fused.py tests/test_gpu_bench_config.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "benchmarked" 2>&1 | tail -4
python -m pytest tests/test_gpu_reference_schedule.py -q -m gpu -k "lockstep_fused and forced" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_view'], d['config']['binning_rounds'])"; done
PROF_STEPS=20 PROF_WARMUP=5 tools/prof.sh r03_b --no-extras --no-pmc 2>&1 | head -22
