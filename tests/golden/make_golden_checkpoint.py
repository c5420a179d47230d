#!/usr/bin/env python
"""G9: tests/golden/checkpoint.npz -- the reference's `GaussianModel.capture()` (scene/gaussian_model.py:61-75) run on a
seeded model with two Adam steps behind it, by IMPORTING the reference's Python under the CPU shim of make_golden.py.
Stored: the seeded parameters, the Adam moments the reference's optimiser reached, the statistics, and the STRUCTURE of
the tuple (element kinds and shapes, state_dict keys in order, param-group names / learning rates / key order) as a JSON
string.  Also: the same model after one MORE reference step resumed through `restore()` (scene/gaussian_model.py:77-93)
-- what "resume = uninterrupted" must reproduce.  Only data leaves this script.
Re-run with:  python tests/golden/make_golden_checkpoint.py"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, CudaToCpu, install_shim  # noqa: E402
from make_golden_densify import NAMES, build  # noqa: E402

ARGS = dict(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
            position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)


def kind(x):
    if torch.is_tensor(x):
        return ["Parameter" if isinstance(x, torch.nn.Parameter) else "Tensor", list(x.shape), str(x.dtype)]
    return [type(x).__name__, x if isinstance(x, (int, float)) else None]


def main():
    install_shim({})
    torch.nn.Module.cuda = lambda self, *a, **k: self
    with CudaToCpu():
        from scene.gaussian_model import GaussianModel
        g = torch.Generator().manual_seed(77)
        P = 257
        gm = build(GaussianModel, g, P)
        gm.active_sh_degree = 1
        tup = gm.capture()
        out = {}
        sd = tup[10]
        structure = {
            "len": len(tup),
            "elements": [kind(x) if i != 10 else ["dict", list(x.keys())] for i, x in enumerate(tup)],
            "state_keys": [int(k) for k in sd["state"].keys()],
            "state_entry_keys": list(sd["state"][0].keys()),
            "state_step": [kind(sd["state"][k]["step"]) + [float(sd["state"][k]["step"])] for k in sd["state"]],
            "group_keys": list(sd["param_groups"][0].keys()),
            "group_names": [gr["name"] for gr in sd["param_groups"]],
            "group_lrs": [gr["lr"] for gr in sd["param_groups"]],
            "group_params": [gr["params"] for gr in sd["param_groups"]],
            "group_eps": sd["param_groups"][0]["eps"], "group_betas": list(sd["param_groups"][0]["betas"]),
        }
        out["structure_json"] = np.array(json.dumps(structure))
        groups = {gr["name"]: gr for gr in gm.optimizer.param_groups}
        for i, n in enumerate(NAMES):
            p = groups[n]["params"][0]
            out[f"p_{n}"] = p.detach().numpy().copy()
            out[f"m_{n}"] = sd["state"][i]["exp_avg"].numpy().copy()
            out[f"v_{n}"] = sd["state"][i]["exp_avg_sq"].numpy().copy()
        out["max_radii2D"] = tup[7].numpy().copy()
        out["xyz_gradient_accum"] = tup[8].numpy().copy()
        out["denom"] = tup[9].numpy().copy()
        out["spatial_lr_scale"] = np.array(float(tup[11]))
        # resume through the reference's restore() into a FRESH model and take one more step with seeded gradients
        tup = tuple(x.detach().clone() if torch.is_tensor(x) else x for x in tup[:10]) + (tup[10], tup[11])
        gm2 = GaussianModel(1)
        gm2.restore(tuple(torch.nn.Parameter(x) if 1 <= i <= 6 else x for i, x in enumerate(tup)),
                    types.SimpleNamespace(**ARGS))
        g2 = torch.Generator().manual_seed(78)
        groups2 = {gr["name"]: gr for gr in gm2.optimizer.param_groups}
        for n in NAMES:
            p = groups2[n]["params"][0]
            grad = 1e-3 * torch.randn(p.shape, generator=g2)
            out[f"g_{n}"] = grad.numpy().copy()
            p.grad = grad
        gm2.optimizer.step()
        for n in NAMES:
            out[f"after_{n}"] = groups2[n]["params"][0].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "checkpoint.npz"), **out)
    print("wrote checkpoint.npz", {k: getattr(v, "shape", None) for k, v in list(out.items())[:6]})
    print(json.dumps(structure)[:600])


if __name__ == "__main__":
    main()
