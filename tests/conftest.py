import os
import sys

import pytest

# The CPU oracle (oracle/tile_ref.c, OpenMP over all host cores) shares the GPU box's host with other jobs: threads that
# spin at the end of a parallel region while the box is oversubscribed (load average 100+ seen) turned a 335 s suite into
# 1900 s once.  Sleeping waiters cost nothing on a quiet host (the 500k legs of test_gpu_fullsize_oracle.py, one lease with a
# load average of 28: 22-29 s passive, 65 s with OMP_WAIT_POLICY=ACTIVE).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when collected on a box without a HIP device
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
