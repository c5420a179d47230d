#!/usr/bin/env python
"""Two-round binning on the bench workload: N1, N2 and the number of unfinished tiles per view, at the initial model
and after k optimiser steps with the bench's synthetic gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.debug import state_views
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.step import ShardedAdam, ViewShardedStep
P, W, H = 1_000_000, 800, 600
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
model = synth.synth_model(P, seed=0, device="cuda", width=W, height=H)
model.init_densification_stats()
pairs = synth.synth_view_set(W, H, device="cuda")
bg = torch.zeros(3, device="cuda")
pix = [synth.synth_pixel_grads(W, H, seed=7 * i, device="cuda") for i in range(3)]
pix2 = [synth.synth_pixel_grads(W, H, seed=100 + 7 * i, device="cuda")[0] for i in range(3)]
opt = ShardedAdam(model.parameters(), [1.6e-4, 2.5e-3, 2.5e-3 / 20, 5e-3, 1e-3, 0.05], eps=1e-15)
fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False, seg1_fraction=frac)
st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, overflow_check_every=0)
fn = lambda i, p, s: [(p["render"], pix[i][0]), (p["rendered_depth"], pix[i][1]), (p["rendered_alpha"], pix[i][2]), (s["render"], pix2[i])]
for it in range(0, 31):
    st.step(pair_grad_fn=fn)
    if it in (0, 5, 10, 20, 30):
        torch.cuda.synchronize()
        rows = []
        for s in fr.slots[:2]:
            v = state_views(P, W, H, s.capacity, s.geom, s.binning, s.img)
            r2 = v["ranges2"]
            rows.append((int(v["counts"][0]), int(v["counts"][2]), int((r2[:, 1] > r2[:, 0]).sum())))
        print("step", it, "views 0,1: (N1, N2, tiles with a segment 2):", rows, "opacity mean %.3f" % float(model.get_opacity.mean()))
