#!/usr/bin/env python
"""How many tile instances of the tight (alpha-footprint AABB) binning would an exact ellipse-vs-tile test
remove?  (decides whether exact culling at emission is worth its bookkeeping)"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.debug import state_views
from binocular3dgs_amd.fused import FusedRasterizer
P, W, H = 1_000_000, 800, 600
dev = "cuda"
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
cam = synth.synth_cameras(W, H, yaws=(0.0,), device=dev)[0]
bg = torch.zeros(3, device=dev)
fr = FusedRasterizer(model, W, H, num_slots=1)
with torch.no_grad():
    fr.render(cam, bg, slot=0)
torch.cuda.synchronize()
N = fr.num_rendered()[0]
s = fr.slots[0]
v = state_views(P, W, H, N, s.geom, s.binning, s.img)
pl, tid = v["point_list"].long(), v["tile_ids"].long()
rec = v["records"][pl]
mx, my, cxx, cxy, cyy, op = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]
gx = (W + 15) // 16
tx, ty = (tid % gx).float() * 16, (tid // gx).float() * 16
tau = torch.log(255.0 * op)
def fmin_rect(size):
    ax, bx, ay, by = tx - mx, tx - mx + size, ty - my, ty - my + size
    inside = (ax <= 0) & (bx >= 0) & (ay <= 0) & (by >= 0)
    best = torch.full_like(mx, 3e38)
    for dx in (ax, bx):
        dy = torch.minimum(torch.maximum(-cxy * dx / cyy, ay), by)
        best = torch.minimum(best, 0.5 * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy)
    for dy in (ay, by):
        dx = torch.minimum(torch.maximum(-cxy * dy / cxx, ax), bx)
        best = torch.minimum(best, 0.5 * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy)
    return inside | (best <= tau)
keep = fmin_rect(15.0)
print("instances N = %d, exact ellipse-tile test keeps %d (%.1f%%)" % (N, int(keep.sum()), 100.0 * keep.float().mean()))
per_g = torch.bincount(pl, minlength=P)
print("tiles per visible Gaussian: mean %.2f, p50 %d, p90 %d, p99 %d, max %d" % (
    per_g[per_g > 0].float().mean(), *[int(torch.quantile(per_g[per_g > 0].float(), q)) for q in (0.5, 0.9, 0.99)], int(per_g.max())))
big = per_g > 64
print("Gaussians with > 64 tiles: %d holding %.1f%% of the instances" % (int(big.sum()), 100.0 * per_g[big].sum() / N))
