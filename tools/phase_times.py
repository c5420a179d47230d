#!/usr/bin/env python
"""Forward-phase vs whole-iteration time of the fused path (1M Gaussians, 800x600, 6 views), graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.step import ViewShardedStep
P, W, H = 1_000_000, 800, 600
dev = "cuda"
pairs = synth.synth_view_set(W, H, device=dev)
bg = torch.zeros(3, device=dev)
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device=dev)
views = []
for i, (c, s, t) in enumerate(pairs):
    views += [(c, 2 * i), (s, 2 * i + 1)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    sg = torch.cuda.Stream()
    sg.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sg):
        fn()
    torch.cuda.current_stream().wait_stream(sg)
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for conc in ("batched", "streams", "serial"):
    model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
    fr = FusedRasterizer(model, W, H, num_slots=6, schedule=conc)
    st = ViewShardedStep(model, pairs, bg, fused=fr)

    def fwd_only():
        with torch.no_grad():
            fr.render_batch(views, bg)

    def grad_fn(i, pkg, spkg):
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]

    tf = timed(fwd_only)
    ts = timed(lambda: st.compute_grads(pair_grad_fn=grad_fn))
    print("%s: forward phase %.3f ms, forward+backward %.3f ms -> backward phase %.3f ms" %
          (conc, tf, ts, ts - tf))
