#!/usr/bin/env python
"""G10: tests/golden/functions.npz -- the reference's loss functions and per-iteration model methods ONE BY ONE, each
called by itself on seeded inputs by IMPORTING the reference's Python under the CPU shim of make_golden.py: value AND the
gradient of EVERY input (G6 pins the same functions only as the composed block of train.py:130-148).

    utils/loss_utils.py:18-21      l1_loss(x, y)  /  l1_loss(x, y, mask) with a [B,1,H,W] mask of arbitrary values
    utils/loss_utils.py:36-66      ssim(img1, img2) on [3,H,W];  ssim(..., size_average=False) on [2,3,H,W]
    utils/loss_utils.py:68-91      SmoothLoss().forward(disparity, image)
    utils/graphics_utils.py:80-125 inverse_warp_images(image, disparity, rows, cols), disparities that leave the image
    scene/gaussian_model.py:149-175 training_setup -> param_groups (names, order, lrs), update_learning_rate
    scene/gaussian_model.py:409-411 add_densification_stats on a seeded gradient / mask
    train.py:196-198               three optimizer.step() calls of the reference's torch.optim.Adam(eps=1e-15)
Only data leaves this script.  Re-run with:  python tests/golden/make_golden_functions.py"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, CudaToCpu, install_shim  # noqa: E402

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def main():
    install_shim({})
    torch.nn.Module.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(97531)
    r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    out = {}
    with CudaToCpu():
        from scene.gaussian_model import GaussianModel
        from utils.graphics_utils import inverse_warp_images
        from utils.loss_utils import SmoothLoss, l1_loss, ssim
        H, W = 37, 45

        # ---- l1_loss ----
        x, y = r(3, H, W).requires_grad_(True), r(3, H, W).requires_grad_(True)
        v = l1_loss(x, y)
        (1.7 * v).backward()
        out.update(l1_x=x.detach().numpy(), l1_y=y.detach().numpy(), l1_val=v.detach().numpy(), l1_gx=x.grad.numpy(),
                   l1_gy=y.grad.numpy())
        x, y = r(2, 3, H, W).requires_grad_(True), r(2, 3, H, W).requires_grad_(True)
        m = (0.2 + r(2, 1, H, W) * (r(2, 1, H, W) > 0.3)).requires_grad_(True)
        v = l1_loss(x, y, mask=m)
        (0.6 * v).backward()
        out.update(l1m_x=x.detach().numpy(), l1m_y=y.detach().numpy(), l1m_m=m.detach().numpy(), l1m_val=v.detach().numpy(),
                   l1m_gx=x.grad.numpy(), l1m_gy=y.grad.numpy(), l1m_gm=m.grad.numpy())

        # ---- ssim ----
        a = r(3, H, W).requires_grad_(True)
        b = (a.detach() + 0.25 * (r(3, H, W) - 0.5)).clamp(0, 1).requires_grad_(True)
        v = ssim(a, b)
        (1.3 * v).backward()
        out.update(ss_a=a.detach().numpy(), ss_b=b.detach().numpy(), ss_val=v.detach().numpy(), ss_ga=a.grad.numpy(),
                   ss_gb=b.grad.numpy())
        a = r(2, 3, H, W).requires_grad_(True)
        b = (a.detach() + 0.4 * (r(2, 3, H, W) - 0.5)).clamp(0, 1).requires_grad_(True)
        v = ssim(a, b, size_average=False)
        wgt = torch.tensor([0.7, -1.9])
        (v * wgt).sum().backward()
        out.update(ssb_a=a.detach().numpy(), ssb_b=b.detach().numpy(), ssb_val=v.detach().numpy(), ssb_w=wgt.numpy(),
                   ssb_ga=a.grad.numpy(), ssb_gb=b.grad.numpy())

        # ---- SmoothLoss ----
        sm = SmoothLoss()
        d = (4.0 * r(2, 1, H, W) - 2.0).requires_grad_(True)
        im = r(2, 3, H, W).requires_grad_(True)
        v = sm.forward(disparity=d, image=im)
        (2.2 * v).backward()
        out.update(sm_d=d.detach().numpy(), sm_im=im.detach().numpy(), sm_val=v.detach().numpy(), sm_gd=d.grad.numpy(),
                   sm_gim=im.grad.numpy())

        # ---- inverse_warp_images ----
        rows, cols = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        im = r(2, 3, H, W).requires_grad_(True)
        d = (14.0 * r(2, 1, H, W) - 7.0).requires_grad_(True)       # many samples leave the 45-pixel rows
        o = inverse_warp_images(im, d, rows, cols)
        up = r(2, 3, H, W) - 0.5
        (o * up).sum().backward()
        out.update(iw_im=im.detach().numpy(), iw_d=d.detach().numpy(), iw_out=o.detach().numpy(), iw_up=up.numpy(),
                   iw_gim=im.grad.numpy(), iw_gd=d.grad.numpy())
        ones = torch.ones(2, 1, H, W)
        d2 = d.detach().clone().requires_grad_(True)
        o = inverse_warp_images(ones, d2, rows, cols)                  # train.py:133: the shift mask
        (o * up[:, :1]).sum().backward()
        out.update(iwm_out=o.detach().numpy(), iwm_gd=d2.grad.numpy())

        # ---- GaussianModel: training_setup / update_learning_rate / add_densification_stats / optimizer.step ----
        P = 203
        gm = GaussianModel(1)
        nn = torch.nn
        gm._xyz = nn.Parameter(torch.randn(P, 3, generator=g))
        gm._features_dc = nn.Parameter(torch.randn(P, 1, 3, generator=g))
        gm._features_rest = nn.Parameter(0.1 * torch.randn(P, 3, 3, generator=g))
        gm._scaling = nn.Parameter(-3.0 + torch.randn(P, 3, generator=g))
        gm._rotation = nn.Parameter(torch.randn(P, 4, generator=g))
        gm._opacity = nn.Parameter(torch.randn(P, 1, generator=g))
        gm.max_radii2D = torch.zeros(P)
        gm.spatial_lr_scale = 3.5
        args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                     position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3,
                                     opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
        gm.training_setup(args)
        out["ts_names"] = np.array([gr["name"] for gr in gm.optimizer.param_groups])
        out["ts_lrs"] = np.array([gr["lr"] for gr in gm.optimizer.param_groups], dtype=np.float64)
        out["ts_eps_betas"] = np.array([gm.optimizer.param_groups[0]["eps"], *gm.optimizer.param_groups[0]["betas"]])
        out["ts_args"] = np.array([args.percent_dense, args.position_lr_init, args.position_lr_final, args.position_lr_delay_mult,
                                   args.position_lr_max_steps, args.feature_lr, args.opacity_lr, args.scaling_lr,
                                   args.rotation_lr, gm.spatial_lr_scale])
        its = np.array([1, 7, 500, 15000, 30000])
        out["ulr_its"] = its
        out["ulr_vals"] = np.array([gm.update_learning_rate(int(i)) for i in its], dtype=np.float64)
        groups = {gr["name"]: gr for gr in gm.optimizer.param_groups}
        for n in NAMES:
            out[f"opt_p0_{n}"] = groups[n]["params"][0].detach().numpy().copy()
        for k in range(3):
            gm.update_learning_rate(100 * (k + 1))
            for n in NAMES:
                p = groups[n]["params"][0]
                p.grad = 1e-3 * torch.randn(p.shape, generator=g)
                out[f"opt_g{k}_{n}"] = p.grad.numpy().copy()
            gm.optimizer.step()
            gm.optimizer.zero_grad(set_to_none=True)
        for n in NAMES:
            p = groups[n]["params"][0]
            st = gm.optimizer.state[p]
            out[f"opt_p3_{n}"] = p.detach().numpy().copy()
            out[f"opt_m3_{n}"] = st["exp_avg"].numpy().copy()
            out[f"opt_v3_{n}"] = st["exp_avg_sq"].numpy().copy()
        out["opt_step3"] = np.array(float(gm.optimizer.state[groups["xyz"]["params"][0]]["step"]))

        vsp = torch.zeros(P, 3, requires_grad=True)
        for k in range(2):
            vsp.grad = torch.randn(P, 3, generator=g) * 1e-3
            filt = torch.rand(P, generator=g) > 0.45
            out[f"ads_grad{k}"] = vsp.grad.numpy().copy()
            out[f"ads_filter{k}"] = filt.numpy().copy()
            gm.add_densification_stats(vsp, filt)
        out["ads_accum"] = gm.xyz_gradient_accum.numpy().copy()
        out["ads_denom"] = gm.denom.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "functions.npz"), **out)
    print("functions.npz", os.path.getsize(os.path.join(OUT, "functions.npz")), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
