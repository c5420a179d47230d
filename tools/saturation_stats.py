#!/usr/bin/env python
"""How much of each tile list does the blend need?  Per tile: did every pixel terminate (T < 1e-4), and how deep in the
global depth order sits the last instance any of its pixels used."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.debug import state_views
P, W, H = 1_000_000, 800, 600
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
model = synth.synth_model(P, seed=0, device="cuda", width=W, height=H, scale_mult=scale)
cam = synth.synth_view_set(W, H, device="cuda")[0][0]
bg = torch.zeros(3, device="cuda")
fr = FusedRasterizer(model, W, H, num_slots=1)
fr.fit_capacity([(cam, 0)], bg)
with torch.no_grad():
    out = fr.render_batch([(cam, 0)], bg)[0]
n = fr.num_rendered()[0]
sl = fr.slots[0]
v = state_views(P, W, H, sl.capacity, sl.geom, sl.binning, sl.img)
pl = v["point_list"][:n].long()
ranges = v["ranges"].long()
ncon = v["n_contrib"].long()
fT = v["final_T"]
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(gy * 16, gx * 16, dtype=torch.long, device="cuda"); pad[:H, :W] = ncon
tmax = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(1).values
padT = torch.zeros(gy * 16, gx * 16, device="cuda"); padT[:H, :W] = fT
# a pixel is "terminated" when the next Gaussian would have pushed T below 1e-4; approximate by final_T < 2e-4/...; use alpha>0.999
unsat = (padT.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256) > 1e-3).any(1)
lens = ranges[:, 1] - ranges[:, 0]
depth = v["depth_bits"].long()
vis = depth[depth != 0xFFFFFFFF if False else slice(None)]
order = torch.argsort(torch.where(v["tiles_touched"] > 0, depth, torch.full_like(depth, 1 << 40)), stable=True)
rank = torch.empty(P, dtype=torch.long, device="cuda"); rank[order] = torch.arange(P, device="cuda")
V = int((v["tiles_touched"] > 0).sum())
ok = lens > 0
last_pos = (ranges[:, 0] + tmax - 1).clamp(min=0)
r_tile = torch.where(tmax > 0, rank[pl[last_pos.clamp(max=n - 1)]], torch.zeros_like(tmax)).float() / V
print(f"scale {scale}: N {n}, V {V}, tiles {gx*gy}, tiles with an unsaturated pixel (final_T > 1e-3): {int((unsat & ok).sum())}")
print("used fraction of the tile lists: mean %.3f" % float((tmax[ok].float() / lens[ok].float()).mean()))
for q in (0.5, 0.9, 0.99, 1.0):
    print(f"  depth-rank fraction of the last needed instance, quantile {q}: %.3f" % float(torch.quantile(r_tile[ok], q)))
for f in (0.2, 0.25, 0.33, 0.5):
    need = (r_tile > f) | unsat
    inst2 = int(lens[need & ok].sum())
    print(f"  segment 1 = nearest {f:.2f} of the visible Gaussians: tiles needing round 2: {int((need & ok).sum())}, their full lists hold {inst2} instances ({inst2 / n:.3f} of N)")
