"""usage (GPU box): python tools/module_surface_profile.py -- device timeline of bench.py's `module_surface` extra (render() as the
reference writes it -> GaussianRasterizer -> the compiled _RasterizeGaussians node -> _C.rasterize_gaussians[_backward], one
node per render, 6 views per iteration at the headline size): kernels by device time, busy vs span, host ops."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import binocular3dgs_amd.render as RM  # noqa: E402

sys.argv = [sys.argv[0]]
args = B.parse()
B.resolve_defaults(args, 1)
args.optimizer = "b3gs"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
RM._FUSED_NODE = False
j = B.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, 6, "weak", path="dropin", graph=False)
j.prepare(3)
el = j.timed_best(10)
print(f"module surface: {10 / el:.1f} iters/s, {el / 10 * 1e3:.3f} ms per iteration")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
N = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for _ in range(N):
        j.eager_step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
evs = sorted((e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")), key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in evs) / N
span = (evs[-1].time_range.end - evs[0].time_range.start) / N
print(f"under the profiler: wall {wall / N * 1e3:.3f} ms/iter; device: {len(evs) / N:.0f} kernels+copies, busy {busy:.0f} us, span {span:.0f} us")
agg = {}
for e in evs:
    a = agg.setdefault(e.name, [0.0, 0])
    a[0] += e.time_range.end - e.time_range.start
    a[1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda r: -r[1][0])[:40]:
    print(f"  dev {t / N:8.1f} us  x{c / N:5.1f}  {k[:130]}")
ka = prof.key_averages()
for k, t, c in sorted(((e.key, e.self_cpu_time_total / N, e.count / N) for e in ka), key=lambda r: -r[1])[:15]:
    print(f"  cpu {t:8.1f} us  x{c:5.1f}  {k[:110]}")
