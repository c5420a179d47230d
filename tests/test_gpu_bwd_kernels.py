"""The three blend-backward kernels of csrc/render.hip produce the same gradients: the default (one wave per tile quadrant,
candidate records through the scalar cache), the same with LDS-staged records (B3GS_BWD_KERNEL=wave) and the
tile-workgroup kernel of rounds 1-2 (B3GS_BWD_KERNEL=tile).  The switch is read once per process, so each kernel runs
in its own interpreter on the same seeded scene; gradients agree to the order of the fp32 atomics (1e-5 relative L2),
the statistics that involve no atomics bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from binocular3dgs_amd import synth
from binocular3dgs_amd.fused import FusedRasterizer
W, H, P = 208, 144, 30000
model = synth.synth_model(P, seed=7, device="cuda", width=W, height=H, K=4)
pairs = synth.synth_view_set(W, H, device="cuda")
bg = torch.tensor([0.1, 0.0, 0.2], device="cuda")
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=3, device="cuda")
views = []
for i, (cam, scam, _t) in enumerate(pairs):
    views += [(cam, 2 * i, True), (scam, 2 * i + 1, False)]
model.init_densification_stats()
fr = FusedRasterizer(model, W, H, num_slots=len(views))
for p in model.parameters():
    p.grad = torch.zeros_like(p)
outs = fr.render_batch(views, bg)
o, g = [], []
for x, v in zip(outs, views):
    o.append(x["render"]); g.append(gc)
    if v[2]:
        o += [x["rendered_depth"], x["rendered_alpha"]]; g += [gd, ga]
torch.autograd.backward(o, g)
torch.cuda.synchronize()
assert not fr.overflowed()
np.savez(sys.argv[1], denom=model.denom.cpu().numpy(), accum=model.xyz_gradient_accum.cpu().numpy(),
         **{"g%%d" %% k: p.grad.cpu().numpy() for k, p in enumerate(model.parameters())})
"""


def _run(tmp_path, which):
    env = dict(os.environ)
    env.pop("B3GS_BWD_KERNEL", None)
    if which:
        env["B3GS_BWD_KERNEL"] = which
    out = tmp_path / f"grads_{which or 'default'}.npz"
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}, str(out)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_the_three_backward_kernels_agree(tmp_path):
    ref = _run(tmp_path, None)
    for which in ("wave", "tile"):
        got = _run(tmp_path, which)
        assert np.array_equal(got["denom"], ref["denom"])
        for k in sorted(ref.files):
            a, b = got[k].astype(np.float64).ravel(), ref[k].astype(np.float64).ravel()
            if b.size == 0:
                continue
            assert np.abs(b).max() > 0, k
            assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b), (which, k)
