#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference's Python (read-only, from
/root/reference) in the build container.  Only data leaves: seeded inputs and the outputs the
reference computed for them.  Re-run with:  python tests/golden/make_golden.py

The reference hard-codes device="cuda"; the shim below stubs the modules
that are absent here and redirects every "cuda" placement to the CPU without touching the
reference's files.  Nothing in this script or in the fixtures contains reference source text.

Fixtures (what each pins):
  G1 sh_eval.npz        utils/sh_utils.py:57-112 eval_sh, degrees 0..3            -> SH->RGB stage
  G2 covariance.npz     scene/gaussian_model.py:27-31,117-118 get_covariance      -> cov3D stage
  G3 cameras.npz        scene/cameras.py:55-58 matrices, utils/graphics_utils.py  -> matrix conventions
  G4 shifted.npz        scene/__init__.py:96-115 getShiftedCamera                 -> binocular view
  G5 render_kwargs.npz  gaussian_renderer/__init__.py:18-103 kwargs routing       -> render() surface
  G6 loss_block.npz     train.py:130-148 + utils/loss_utils.py + inverse_warp     -> pixel gradients
  G7 misc.npz           get_expon_lr_func, inverse_sigmoid, psnr
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def install_shim(record):
    from typing import NamedTuple
    sys.path.insert(0, REF)
    for name in ("plyfile", "simple_knn", "simple_knn._C", "imageio", "skimage", "skimage.transform"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None
    sys.modules["simple_knn"]._C = sys.modules["simple_knn._C"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    class GaussianRasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            record["settings"] = self.rs
            record["kwargs"] = kw
            P = kw["means3D"].shape[0]
            H, W = self.rs.image_height, self.rs.image_width
            return (torch.zeros(3, H, W), torch.zeros(P, dtype=torch.int32), torch.zeros(1, H, W),
                    torch.zeros(1, H, W))

    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings = GaussianRasterizationSettings
    m.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = m


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs and kwargs["device"] is not None and "cuda" in str(kwargs["device"]):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to and len(args) > 1 and isinstance(args[1], (str, torch.device)) and "cuda" in str(args[1]):
            return args[0]
        if func is torch.nn.Module.cuda:
            return args[0]
        return func(*args, **kwargs)


def main():
    record = {}
    install_shim(record)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(1234)
    with CudaToCpu():
        from utils.sh_utils import eval_sh
        from utils.general_utils import get_expon_lr_func, inverse_sigmoid
        from utils.graphics_utils import inverse_warp_images
        from utils.image_utils import psnr
        from utils.loss_utils import SmoothLoss, l1_loss, ssim
        from scene.cameras import Camera
        from scene.gaussian_model import GaussianModel
        from scene import Scene
        from gaussian_renderer import render

        # ---- G1 ----
        P = 257
        sh = torch.randn(P, 3, 16, generator=g)
        dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g))
        out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
        for deg in range(4):
            out[f"rgb_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
        np.savez_compressed(os.path.join(OUT, "sh_eval.npz"), **out)

        # ---- G2 ----
        gm = GaussianModel(1)
        P = 300
        gm._xyz = torch.randn(P, 3, generator=g)
        gm._features_dc = torch.randn(P, 1, 3, generator=g)
        gm._features_rest = torch.randn(P, 3, 3, generator=g) * 0.1
        gm._scaling = math.log(0.05) + 0.7 * torch.randn(P, 3, generator=g)
        gm._rotation = torch.randn(P, 4, generator=g)
        gm._opacity = torch.randn(P, 1, generator=g)
        np.savez_compressed(os.path.join(OUT, "covariance.npz"), scaling_raw=gm._scaling.numpy(),
                            rotation_raw=gm._rotation.numpy(), opacity_raw=gm._opacity.numpy(),
                            cov_mod1=gm.get_covariance(1.0).numpy(), cov_mod07=gm.get_covariance(0.7).numpy(),
                            get_scaling=gm.get_scaling.numpy(), get_rotation=gm.get_rotation.numpy(),
                            get_opacity=gm.get_opacity.numpy(), get_features=gm.get_features.numpy(),
                            features_dc=gm._features_dc.numpy(), features_rest=gm._features_rest.numpy())

        # ---- G3 / G4 ----
        cams = {}
        rng = np.random.default_rng(7)
        specs = []
        for i in range(4):
            A = rng.normal(size=(3, 3))
            Q, _ = np.linalg.qr(A)
            if np.linalg.det(Q) < 0:
                Q[:, 0] *= -1
            T = rng.normal(size=3) * 2.0
            specs.append((Q, T, math.radians(40 + 10 * i), math.radians(30 + 8 * i), 64 + 16 * i, 48 + 8 * i))
        specs.append((np.eye(3), np.zeros(3), math.radians(60.0), 2 * math.atan(math.tan(math.radians(30.0)) * 600 / 800), 800, 600))
        for i, (R, T, fx, fy, w, h) in enumerate(specs):
            cam = Camera(colmap_id=i, R=R, T=T, FoVx=fx, FoVy=fy, image=torch.rand(3, h, w, generator=g),
                         gt_alpha_mask=None, image_name=str(i), uid=i)
            cams[f"R{i}"], cams[f"T{i}"] = R, T
            cams[f"fov{i}"] = np.array([fx, fy, w, h])
            cams[f"wvt{i}"] = cam.world_view_transform.numpy()
            cams[f"proj{i}"] = cam.projection_matrix.numpy()
            cams[f"full{i}"] = cam.full_proj_transform.numpy()
            cams[f"center{i}"] = cam.camera_center.numpy()
            cams[f"focal{i}"] = np.array(cam.get_focal())
            for j, t in enumerate((0.1, -0.1, 0.4, -0.4)):
                sc = Scene.getShiftedCamera(None, cam, t)
                cams[f"shift{i}_{j}_t"] = np.array(t)
                cams[f"shift{i}_{j}_wvt"] = sc.world_view_transform.numpy()
                cams[f"shift{i}_{j}_full"] = sc.full_proj_transform.numpy()
                cams[f"shift{i}_{j}_center"] = sc.camera_center.numpy()
        cams["n"] = np.array(len(specs))
        np.savez_compressed(os.path.join(OUT, "cameras.npz"), **cams)

        # ---- G5: kwargs routing of render() for the 4 pipe-flag combinations ----
        class Pipe:
            def __init__(self, a, b):
                self.convert_SHs_python, self.compute_cov3D_python, self.debug = a, b, False
        R, T, fx, fy, w, h = specs[1]
        cam = Camera(colmap_id=0, R=R, T=T, FoVx=fx, FoVy=fy, image=torch.rand(3, h, w, generator=g),
                     gt_alpha_mask=None, image_name="x", uid=0)
        gm.active_sh_degree = 1
        rk = {"xyz": gm._xyz.numpy(), "camR": R, "camT": T, "camfov": np.array([fx, fy, w, h])}
        for a in (False, True):
            for b in (False, True):
                render(cam, gm, Pipe(a, b), torch.tensor([0.1, 0.2, 0.3]), scaling_modifier=0.9)
                tag = f"sh{int(a)}_cov{int(b)}"
                rs, kw = record["settings"], record["kwargs"]
                rk[f"{tag}_present"] = np.array([kw[k] is not None for k in
                                                 ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales",
                                                  "rotations", "cov3D_precomp")])
                for k in ("shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"):
                    if kw[k] is not None:
                        rk[f"{tag}_{k}"] = kw[k].detach().numpy()
                rk[f"{tag}_settings"] = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy,
                                                  rs.scale_modifier, rs.sh_degree, float(rs.prefiltered), float(rs.debug)])
                rk[f"{tag}_bg"] = rs.bg.numpy()
                rk[f"{tag}_viewmatrix"] = rs.viewmatrix.numpy()
                rk[f"{tag}_projmatrix"] = rs.projmatrix.numpy()
                rk[f"{tag}_campos"] = rs.campos.numpy()
        np.savez_compressed(os.path.join(OUT, "render_kwargs.npz"), **rk)

        # ---- G6: the loss block of train.py:130-148 on seeded images, values and pixel gradients ----
        Hh, Ww = 40, 56
        image = torch.rand(3, Hh, Ww, generator=g).requires_grad_(True)
        depth = (1.0 + 4.0 * torch.rand(1, Hh, Ww, generator=g)).requires_grad_(True)
        alpha = torch.rand(1, Hh, Ww, generator=g).requires_grad_(True)
        shifted = torch.rand(3, Hh, Ww, generator=g).requires_grad_(True)
        gt = torch.rand(3, Hh, Ww, generator=g)
        gt_alpha_mask = (torch.rand(1, Hh, Ww, generator=g) > 0.3).float()
        focal_x, trans_dist, lambda_dssim = 55.0, 0.23, 0.2
        row_indices, column_indices = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
        mask = torch.ones(1, Hh, Ww)
        smooth = SmoothLoss()
        disparity = focal_x * (-trans_dist) / (depth + 1e-5)
        warped = inverse_warp_images(shifted.unsqueeze(0), disparity.unsqueeze(0), row_indices, column_indices)
        shift_mask = inverse_warp_images(mask.unsqueeze(0), disparity.unsqueeze(0), row_indices, column_indices)
        l1_masked = l1_loss(warped, gt.unsqueeze(0), mask=shift_mask)
        sm = smooth.forward(disparity=disparity * shift_mask, image=gt.unsqueeze(0))
        disparity_loss = l1_masked + 0.05 * sm
        alpha_loss = torch.mean(torch.abs(alpha) * (1 - gt_alpha_mask))
        Ll1 = l1_loss(image, gt)
        ss = ssim(image, gt)
        loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ss)
        total = loss + disparity_loss + alpha_loss
        total.backward()
        np.savez_compressed(os.path.join(OUT, "loss_block.npz"), image=image.detach().numpy(),
                            depth=depth.detach().numpy(), alpha=alpha.detach().numpy(),
                            shifted=shifted.detach().numpy(), gt=gt.numpy(), gt_alpha_mask=gt_alpha_mask.numpy(),
                            scalars=np.array([focal_x, trans_dist, lambda_dssim]),
                            warped=warped.detach().numpy(), shift_mask=shift_mask.detach().numpy(),
                            l1_masked=l1_masked.detach().numpy(), smooth=sm.detach().numpy(),
                            alpha_loss=alpha_loss.detach().numpy(), Ll1=Ll1.detach().numpy(), ssim=ss.detach().numpy(),
                            total=total.detach().numpy(), g_image=image.grad.numpy(), g_depth=depth.grad.numpy(),
                            g_alpha=alpha.grad.numpy(), g_shifted=shifted.grad.numpy())

        # ---- G7 ----
        f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
        steps = np.array([0, 1, 10, 100, 1000, 15000, 30000])
        x = torch.rand(50, generator=g) * 0.98 + 0.01
        a, b = torch.rand(3, 20, 30, generator=g), torch.rand(3, 20, 30, generator=g)
        np.savez_compressed(os.path.join(OUT, "misc.npz"), lr_steps=steps, lr_vals=np.array([f(int(s)) for s in steps]),
                            isig_x=x.numpy(), isig_y=inverse_sigmoid(x).numpy(), psnr_a=a.numpy(), psnr_b=b.numpy(),
                            psnr=psnr(a, b).numpy())
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()
