#!/usr/bin/env python
"""Burst vs sustained: replay the iteration graph back to back and print ms/iteration per block of 25, with the
shader clock / power rocm-smi reports in between."""
import os
import subprocess
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
P, W, H = 1_000_000, 800, 600
dev = "cuda"
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
model.init_densification_stats()
pairs = synth.synth_view_set(W, H, device=dev)
bg = torch.zeros(3, device=dev)
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device=dev)
opt = FusedAdam(model.parameters(), [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05], eps=1e-15)
fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False)
st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr)
fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
for _ in range(3):
    st.step(pair_grad_fn=fn)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.step(pair_grad_fn=fn)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st.step(pair_grad_fn=fn)
torch.cuda.synchronize()


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.split("\n") if any(k in ln for k in ("sclk", "Power", "junction", "Socket"))]
        return " | ".join(keep[:6])
    except Exception as e:  # noqa: BLE001
        return repr(e)


print("idle:", smi())
series = []
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    t0 = time.perf_counter()
    for _ in range(25):
        g.replay()
    torch.cuda.synchronize()
    series.append((time.perf_counter() - t0) / 25 * 1e3)
    if blk in (3, 11, 23):
        print("after %d iterations:" % (25 * (blk + 1)), smi())
print("ms/iter per block of 25:", " ".join("%.2f" % x for x in series))
