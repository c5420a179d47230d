"""Where an iteration of a surface of bench_ref_schedule.py goes: device time per kernel, the idle gaps of the device
between kernels (and what follows them), host time per op (torch profiler), wall clock per iteration.
usage: python tools/unchanged_profile.py [P W H] [surface]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ref_schedule as B  # noqa: E402

P, W, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (500_000, 800, 600)
surface = sys.argv[4] if len(sys.argv) > 4 else "unchanged"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
holder = {}
orig = B._count_launches


def grab(fn, steps=3):
    holder["step"] = fn
    return orig(fn, steps)


B._count_launches = grab
res = B.run(dev, P, W, H, 60.0, surface, steps=40)
print(res)
step = holder["step"]
from torch.profiler import ProfilerActivity, profile  # noqa: E402
for _ in range(5):
    step()
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"wall under profiler: {el / N * 1e3:.3f} ms/iter")
evs = [e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")]
evs.sort(key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in evs) / N
span = (evs[-1].time_range.end - evs[0].time_range.start) / N
print(f"device: {len(evs) / N:.1f} kernels+copies per iteration, busy {busy:.1f} us, span {span:.1f} us per iteration")
agg = {}
for e in evs:
    a = agg.setdefault(e.name, [0.0, 0])
    a[0] += e.time_range.end - e.time_range.start
    a[1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda r: -r[1][0])[:50]:
    print(f"  dev {t / N:8.1f} us  x{c / N:5.1f}  {k[:120]}")
# idle gaps of the device inside ONE iteration (the last one), in order
k0 = len(evs) - len(evs) // N
print("gaps > 8 us in the last iteration (gap, then the kernel that ends it):")
for a, b in zip(evs[k0 - 1:-1], evs[k0:]):
    gap = b.time_range.start - a.time_range.end
    if gap > 8:
        print(f"  {gap:7.1f} us  after {a.name[:50]:50s} -> {b.name[:60]}")
ka = prof.key_averages()
for k, t, c in sorted(((e.key, e.self_cpu_time_total / N, e.count / N) for e in ka), key=lambda r: -r[1])[:30]:
    print(f"  cpu {t:8.1f} us  x{c:5.1f}  {k[:110]}")
from binocular3dgs_amd import rasterizer as R  # noqa: E402
print("rasterizer stats:", dict(R._stats))
