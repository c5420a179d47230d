#!/usr/bin/env python
"""How fast can the tile lists be written DIRECTLY (each instance to its final position) in emission (depth) order?
Pattern of the real workload: instance j of the depth-ordered emission goes to dest[j] = its position in the tile-major
list.  Measures a plain scattered 4-byte store (torch index_put) against a contiguous copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.debug import state_views
P, W, H = 1_000_000, 800, 600
model = synth.synth_model(P, seed=0, device="cuda", width=W, height=H)
pairs = synth.synth_view_set(W, H, device="cuda")
bg = torch.zeros(3, device="cuda")
fr = FusedRasterizer(model, W, H, num_slots=1)
fr.fit_capacity([(pairs[0][0], 0)], bg)
with torch.no_grad():
    fr.render_batch([(pairs[0][0], 0)], bg)
n = fr.num_rendered()[0]
sl = fr.slots[0]
v = state_views(P, W, H, sl.capacity, sl.geom, sl.binning, sl.img)
pl, tid = v["point_list"][:n].long(), v["tile_ids"][:n].long()
depth = v["depth_bits"].long()[pl]
# emission order = (depth, gaussian index, tile): sort the final list back into that order
key = (depth * (1 << 21) + pl) * 2048 + tid
order = torch.argsort(key)            # order[j] = final position of the j-th emitted instance
dest = order.int()
src = torch.arange(n, device="cuda", dtype=torch.int32)
outs = [torch.empty(n, device="cuda", dtype=torch.int32) for _ in range(6)]
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
d64 = dest.long()
print("n", n)
print("scatter 1 view  (index_put, 4B): %.1f us" % t(lambda: outs[0].index_put_((d64,), src)))
print("scatter 6 views sequential:      %.1f us" % t(lambda: [o.index_put_((d64,), src) for o in outs]))
print("contiguous copy 1 view:          %.1f us" % t(lambda: outs[0].copy_(src)))
big_d = torch.cat([d64 + k * n for k in range(6)]); big_s = src.repeat(6); big_o = torch.empty(6 * n, device="cuda", dtype=torch.int32)
print("scatter 6 views one launch:      %.1f us" % t(lambda: big_o.index_put_((big_d,), big_s)))
