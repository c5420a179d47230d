#!/usr/bin/env python
"""What does the loss block (train.py:123-148) cost next to the rasterizer?  Times one iteration (3 pairs,
1M Gaussians, 800x600, HIP graph) with synthetic pixel gradients and with the real loss block."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.loss import binocular_loss
from binocular3dgs_amd.step import ViewShardedStep
P, W, H = 1_000_000, 800, 600
dev = "cuda"
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
pairs = synth.synth_view_set(W, H, device=dev)
bg = torch.zeros(3, device=dev)
gts = [torch.rand(3, H, W, device=dev) for _ in pairs]
masks = [(g.max(0, keepdim=True).values < 0.1).float() for g in gts]
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device=dev)
model.init_densification_stats()
fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False)
st = ViewShardedStep(model, pairs, bg, fused=fr)
USE_FUSED_LOSS = len(sys.argv) > 1 and sys.argv[1] == "fused"


def loss_fn(i, cam, pkg, spkg, t):
    if USE_FUSED_LOSS:
        from binocular3dgs_amd.fused_loss import binocular_loss_fused
        return binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i],
                                    shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t,
                                    bg_mask=masks[i])
    return binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i], shifted_image=spkg["render"],
                          focal_x=cam.get_focal()[0], trans_dist=t, bg_mask=masks[i])[0]


def grad_fn(i, pkg, spkg):
    return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = timed(lambda: st.compute_grads(pair_grad_fn=grad_fn))
b = timed(lambda: st.compute_grads(loss_fn=loss_fn))
print("iteration with synthetic pixel gradients %.3f ms, with the loss block %.3f ms -> loss block %.3f ms (%s)" %
      (a, b, b - a, "fused HIP" if USE_FUSED_LOSS else "PyTorch ops"))
