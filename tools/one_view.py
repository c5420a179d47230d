#!/usr/bin/env python
"""Single-view forward+backward driver through the drop-in render(): target for rocprofv3 --pmc
passes (tools/pmc.sh) and quick stage timing.  usage: tools/one_view.py [P] [W] [H] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from binocular3dgs_amd import synth  # noqa: E402
from binocular3dgs_amd.render import PipelineParams, render  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 600
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = "cuda"
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
cam = synth.synth_cameras(W, H, yaws=(0.0,), device=dev)[0]
bg = torch.zeros(3, device=dev)
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device=dev)
for _ in range(reps):
    pkg = render(cam, model, PipelineParams(), bg)
    torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
torch.cuda.synchronize()
print("ok", int((pkg["radii"] > 0).sum()))
