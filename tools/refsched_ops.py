"""usage (GPU box): python tools/refsched_ops.py [fused|render] -- torch-profiler table of the reference's two-view iteration
(bench_ref_schedule.py) on one surface: which aten ops / copies / fills surround the HIP kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_ref_schedule as B
surface = sys.argv[1] if len(sys.argv) > 1 else "fused"
from torch.profiler import profile, ProfilerActivity
_orig = B._count_launches


def patched(fn, steps=3):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
    return 0.0


B._count_launches = patched
print(B.run(torch.device("cuda", 0), 500000, 800, 600, 60.0, surface, steps=10, warmup=40))
