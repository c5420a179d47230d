#!/bin/bash
# usage (GPU box, repo root): tools/switch_matrix_m.sh -- the fused path's parity tests under each measurement (A/B) switch of DESIGN 5.3
T="tests/test_gpu_fused.py tests/test_gpu_bench_config.py"
for sw in "" B3GS_NO_KEY27=1 B3GS_NO_LPT=1 B3GS_NO_FWD_LPT=1 B3GS_ROUND2_LEGACY=1 B3GS_BWD_KERNEL=tile B3GS_BWD_KERNEL=wave B3GS_ACC_PER_THREAD=1 B3GS_ACC_PER_THREAD=4 B3GS_SORT9_ITEMS=8 B3GS_SORT9_ITEMS=16 B3GS_SEG1_FRAC=0 B3GS_SEG1_FRAC=0.25 B3GS_BWD_VIEW_GROUPS=1 B3GS_BWD_VIEW_GROUPS=6; do
  r=$(env $sw python -m pytest $T -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tr "\n" " ")
  echo "${sw:-defaults}: $r"
done
