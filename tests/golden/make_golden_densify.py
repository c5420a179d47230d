#!/usr/bin/env python
"""G8: tests/golden/densify.npz -- the reference's densify_and_prune (scene/gaussian_model.py:258-407: clone,
split, prune, Adam-state surgery) and opacity_decay (:307-309) run on seeded inputs by IMPORTING the reference's
Python under the CPU shim of make_golden.py.  torch.normal is replaced by `mean + std * injected_noise` for the
duration of the call so that the random split offsets are part of the fixture.  Only data leaves this script.
Re-run with:  python tests/golden/make_golden_densify.py"""
import math
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, CudaToCpu, install_shim  # noqa: E402

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # the reference's param-group order


def build(GaussianModel, g, P):
    gm = GaussianModel(1)
    nn = torch.nn
    gm._xyz = nn.Parameter(torch.randn(P, 3, generator=g))
    gm._features_dc = nn.Parameter(torch.randn(P, 1, 3, generator=g))
    gm._features_rest = nn.Parameter(0.1 * torch.randn(P, 3, 3, generator=g))
    gm._scaling = nn.Parameter(math.log(0.05) + 1.2 * torch.randn(P, 3, generator=g))
    gm._rotation = nn.Parameter(torch.randn(P, 4, generator=g))
    gm._opacity = nn.Parameter(2.5 * torch.randn(P, 1, generator=g) - 2.0)
    gm.max_radii2D = torch.zeros(P)
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3,
                                 opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
    gm.spatial_lr_scale = 1.0
    gm.training_setup(args)
    for _ in range(2):   # non-trivial Adam state
        for p in (gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation):
            p.grad = 1e-3 * torch.randn(p.shape, generator=g)
        gm.optimizer.step()
    gm.xyz_gradient_accum = (4e-4 * torch.rand(P, 1, generator=g)) * (torch.rand(P, 1, generator=g) > 0.3)
    gm.denom = torch.randint(0, 3, (P, 1), generator=g).float()          # zeros -> 0/0 = NaN -> 0 and x/0 = inf
    gm.max_radii2D = 40.0 * torch.rand(P, generator=g)
    return gm


def snapshot(gm, prefix):
    out = {}
    groups = {gr["name"]: gr for gr in gm.optimizer.param_groups}
    for n in NAMES:
        p = groups[n]["params"][0]
        st = gm.optimizer.state.get(p, None)
        out[f"{prefix}{n}"] = p.detach().numpy().copy()
        out[f"{prefix}{n}_exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"{prefix}{n}_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    return out


def main():
    install_shim({})
    torch.nn.Module.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(4321)
    out = {}
    with CudaToCpu():
        from scene.gaussian_model import GaussianModel
        for case, (extent, size_thr) in {"a": (3.0, None), "b": (1.2, 20)}.items():
            P = 400
            gm = build(GaussianModel, g, P)
            out.update(snapshot(gm, f"{case}_in_"))
            out[f"{case}_accum"] = gm.xyz_gradient_accum.numpy().copy()
            out[f"{case}_denom"] = gm.denom.numpy().copy()
            out[f"{case}_max_radii2D"] = gm.max_radii2D.numpy().copy()
            noise = torch.randn(2 * P, 3, generator=g)
            used = {}

            def fake_normal(mean=None, std=None, **kw):
                n = mean.shape[0]
                used["n"] = n
                return mean + std * noise[:n]
            real = torch.normal
            torch.normal = fake_normal
            try:
                gm.densify_and_prune(2e-4, 0.005, extent, size_thr)
            finally:
                torch.normal = real
            out[f"{case}_noise"] = noise[:used.get("n", 0)].numpy().copy()   # rows: (k, j-th selected) k-major
            out[f"{case}_scalars"] = np.array([2e-4, 0.005, extent, 0.01, -1.0 if size_thr is None else size_thr])
            out.update(snapshot(gm, f"{case}_out_"))
            out[f"{case}_out_stats"] = np.stack([gm.xyz_gradient_accum.numpy()[:, 0], gm.denom.numpy()[:, 0], gm.max_radii2D.numpy()])
        # opacity decay
        gm = build(GaussianModel, g, 64)
        out["decay_in"] = gm._opacity.detach().numpy().copy()
        gm.opacity_decay(factor=0.995)
        out["decay_out"] = gm._opacity.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "densify.npz"), **out)
    for k in ("a", "b"):
        print(k, "P:", out[f"{k}_in_xyz"].shape[0], "->", out[f"{k}_out_xyz"].shape[0], "split samples:", out[f"{k}_noise"].shape[0])
    print("densify.npz", os.path.getsize(os.path.join(OUT, "densify.npz")), "bytes")


if __name__ == "__main__":
    main()
