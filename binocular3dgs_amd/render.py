"""render(): the drop-in for gaussian_renderer/__init__.py:18-103.

Same signature, same flag routing (`pipe.convert_SHs_python`, `pipe.compute_cov3D_python`,
`pipe.debug`, `override_color`, `scaling_modifier`), same returned dict keys/shapes/dtypes.
"""
from __future__ import annotations

import math
import os

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_raw, raw_model_ok, viewspace_leaf

_HipRasterizer = GaussianRasterizer                            # (tests swap `GaussianRasterizer` for recording stubs)
_FUSED_NODE = os.environ.get("B3GS_DROPIN_FUSED", "1") != "0"  # render() of a raw-parameter model as ONE autograd node

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH -> colour on the PyTorch side (the `--convert_SHs_python` path; utils/sh_utils.py:57-112,
    degrees 0..3).  sh: [..., C, K], dirs: [..., 3] unit vectors -> [..., C]."""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    result = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] +
                      _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + _C2[3] * xz * sh[..., 7] +
                      _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10] +
                          _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] +
                          _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14] +
                          _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


class PipelineParams:
    """arguments/__init__.py:65-70"""

    def __init__(self, convert_SHs_python=False, compute_cov3D_python=False, debug=False):
        self.convert_SHs_python = convert_SHs_python
        self.compute_cov3D_python = compute_cov3D_python
        self.debug = debug


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene; `bg_color` must live on the GPU (as in the reference)."""
    cam = viewpoint_camera
    xyz = pc.get_xyz
    # A model that keeps the reference's raw parameters and activations, default pipeline flags: one autograd node on the
    # raw tensors (activations in-kernel, rasterizer.rasterize_raw) instead of ~30 PyTorch kernels around the rasterizer
    # call.  B3GS_DROPIN_FUSED=0 keeps the statement-by-statement path below.
    fused_node = (_FUSED_NODE and override_color is None and not pipe.compute_cov3D_python and not pipe.convert_SHs_python
                  and GaussianRasterizer is _HipRasterizer and xyz.shape[0] > 0 and raw_model_ok(pc))
    if fused_node:
        # (a leaf instead of the reference's `zeros + 0`: its .grad then IS the node's screen-space gradient tensor,
        # where retain_grad() on a non-leaf clones it -- 12 B per Gaussian per render; all renders of a (device, P) share
        # one block of zeros behind their leaves: nobody reads or writes the values)
        screen_xy = viewspace_leaf(xyz)
    else:
        # zero tensor whose .grad receives the screen-space mean gradients (densification statistic)
        screen_xy = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
        try:
            screen_xy.retain_grad()
        except Exception:
            pass

    settings = GaussianRasterizationSettings(
        image_height=int(cam.image_height),
        image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=cam.camera_center,
        prefiltered=False,
        debug=pipe.debug,
    )
    if fused_node:
        # (the forward of a render that will be differentiated may still be pending when this returns: rasterizer._LazyOut)
        image, radii, depth, alpha, visible = rasterize_raw(pc, screen_xy, settings, cam, lazy_outputs=True)
        return {"render": image,
                "viewspace_points": screen_xy,
                "visibility_filter": visible,
                "radii": radii,
                "rendered_depth": depth,
                "rendered_alpha": alpha}

    raster = GaussianRasterizer(raster_settings=settings)
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            sh_cols = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            view_dir = xyz - cam.camera_center.repeat(feats.shape[0], 1)
            unit_dir = view_dir / view_dir.norm(dim=1, keepdim=True)
            rgb = eval_sh(pc.active_sh_degree, sh_cols, unit_dir)
            colors_precomp = torch.clamp_min(rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    image, radii, depth, alpha = raster(
        means3D=xyz, means2D=screen_xy, shs=shs, colors_precomp=colors_precomp,
        opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    return {"render": image,
            "viewspace_points": screen_xy,
            "visibility_filter": radii > 0,
            "radii": radii,
            "rendered_depth": depth,
            "rendered_alpha": alpha}
