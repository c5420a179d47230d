"""usage (GPU box): python tools/unchanged_host_profile.py -- cProfile of the `unchanged` surface's iteration
(bench_ref_schedule.py) on a scene small enough for the device to be idle: where the HOST time goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ref_schedule as B  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
holder = {}
B._count_launches = lambda fn, steps=3: holder.setdefault("step", fn) and 0
res = B.run(dev, 3000, 128, 96, 60.0, sys.argv[1] if len(sys.argv) > 1 else "unchanged", steps=40)
print(res)
step = holder["step"]
for _ in range(20):
    step()
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
N = 300
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print(f"per iteration: {st.total_tt / N * 1e6:.1f} us of profiled host time")
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(30)
