"""usage (GPU box): python tools/dropin_ops.py -- torch-profiler table of one drop-in iteration (render() per view + torch
autograd): which aten ops / copies / fills surround the HIP kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch, bench
sys.argv = sys.argv[:1] + ["--path", "dropin", "--optimizer", "b3gs", "--graph", "0"]
args = bench.parse()
dev = torch.device("cuda", 0)
job = bench.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, 6, "weak", path="dropin", graph=False)
for _ in range(3):
    job.eager_step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        job.eager_step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
