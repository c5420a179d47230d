#!/bin/bash
# usage (GPU box, repo root): tools/switch_matrix.sh -- the semantic tests of the zero-change surface (lock-step with the
# oracle-backed CPU trainer, the default node against the oracle at 500k) under every product switch of DESIGN 5.3
T="tests/test_gpu_reference_schedule.py::test_reference_schedule_lockstep_hip_vs_oracle_backed_cpu tests/test_gpu_reference_schedule.py::test_reference_schedule_lockstep_swapped_imports_vs_oracle_backed_cpu tests/test_gpu_fullsize_oracle.py::test_default_render_node_vs_oracle"
for sw in "" B3GS_DROPIN_LAZY=0 B3GS_DROPIN_FUSED=0 B3GS_DROPIN_LAZY_MAX=6 B3GS_DROPIN_LAZY_IDLE=0 B3GS_DROPIN_ORDER_HINT=0 B3GS_DROPIN_INPLACE_GRADS=0 B3GS_DROPIN_SYNC=1 B3GS_NO_STAGED=1; do
  r=$(env $sw python -m pytest $T -q -m gpu -x -k "not 1M" 2>&1 | tail -1)
  echo "${sw:-defaults}: $r"
done
