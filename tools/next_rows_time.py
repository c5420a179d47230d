#!/usr/bin/env python
"""Timings of the 8f rows on the device: k-NN init and densify/prune at 1M Gaussians."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.densify import densify_and_prune
from binocular3dgs_amd.init_points import knn_mean_dist2
from binocular3dgs_amd.step import FusedAdam
P = 1_000_000
model = synth.synth_model(P, seed=0, device="cuda", width=800, height=600)
model.init_densification_stats()
for n in (100_000, P):
    pts = model.get_xyz.detach()[:n].contiguous()
    knn_mean_dist2(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    knn_mean_dist2(pts)
    torch.cuda.synchronize()
    print("knn_mean_dist2 P=%d: %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
opt = FusedAdam(model.parameters(), [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05], eps=1e-15)
g = torch.Generator(device="cuda").manual_seed(0)
model.xyz_gradient_accum = 4e-4 * torch.rand(P, 1, device="cuda", generator=g)
model.denom = torch.ones(P, 1, device="cuda")
for rep in range(3):
    P0 = model.get_xyz.shape[0]
    model.xyz_gradient_accum = 4e-4 * torch.rand(P0, 1, device="cuda", generator=g) * (torch.rand(P0, 1, device="cuda", generator=g) < 0.3)
    model.denom = torch.ones(P0, 1, device="cuda")
    with torch.no_grad():
        model._opacity.data[torch.rand(P0, device="cuda", generator=g) < 0.2] = -8.0     # some get pruned
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    newP = densify_and_prune(model, opt, 2e-4, 0.005, 5.0, generator=g)
    torch.cuda.synchronize()
    print("densify_and_prune P=%d -> %d: %.1f ms (incl. the one host read-back and the allocations)" % (P0, newP, (time.perf_counter() - t0) * 1e3))
