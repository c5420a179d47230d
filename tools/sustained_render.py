"""usage (GPU box): python tools/sustained_render.py -- 600 iterations of the reference's two-view schedule through render()
(bench_ref_schedule surfaces "render" and "unchanged") at 300k Gaussians: iters/s, allocator state, what is still pending
and the raw node's counters afterwards (nothing may pile up)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench_ref_schedule as B
from binocular3dgs_amd import rasterizer as R
dev = torch.device("cuda", 0)
for surf in ("render", "unchanged"):
    r = B.run(dev, 300000, 504, 378, 60.0, surf, steps=600, warmup=8)
    torch.cuda.synchronize()
    print(surf, r["iters_per_s"], "alloc MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20,
          "pending tokens", len(R._lazy.pending), "pending fwd", {k: len(v) for k, v in R._pending_fwd.items()}, dict(R._stats))
