#!/bin/bash
# repair-kernel validation: two-round tests, then A/B bench
python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "two_round or fused_equals or sparse" 2>&1 | tail -5
echo "== one round"; tools/ab_env.sh B3GS_SEG1_FRAC "0.125" 2>&1 | tail -4
echo "== legacy two round"; B3GS_ROUND2_LEGACY=1 B3GS_SEG1_FRAC=0.125 python bench.py --steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('legacy', d['value'], d.get('stage_ms_per_view'))"
B3GS_SEG1_FRAC=0.125 PROF_STEPS=10 PROF_WARMUP=3 tools/prof.sh r03_a_two_round --no-extras --no-pmc 2>&1 | head -30
