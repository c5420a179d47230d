"""Shared scene builders for the tests (seeded, CPU tensors)."""
import math

import numpy as np
import torch

from binocular3dgs_amd import synth
from binocular3dgs_amd.camera import Camera, look_at_orbit


def small_scene(P=400, W=64, H=48, seed=0, K=4, sh_degree=1, fovx_deg=60.0, yaw=5.0, scale_mu=0.06, near_frac=0.0):
    """Random Gaussians in front of a slightly rotated camera; activated parameters (what the
    rasterizer receives).  near_frac of the points are pushed to / behind the near plane."""
    p = synth.synth_gaussians(P, seed, W, H, fovx_deg, K)
    g = torch.Generator().manual_seed(seed + 77)
    scaling = math.log(scale_mu) + 0.6 * torch.randn(P, 3, generator=g)
    xyz = p["xyz"].clone()
    if near_frac > 0:
        n = int(P * near_frac)
        xyz[:n, 2] = 0.4 * torch.rand(n, generator=g) - 0.1
    fovx = math.radians(fovx_deg)
    fovy = synth.fovy_from(fovx, W, H)
    R, T = look_at_orbit(yaw)
    cam = Camera(R, T, fovx, fovy, W, H)
    d = dict(
        means3D=xyz, opacities=torch.sigmoid(p["opacity"]), scales=torch.exp(scaling),
        rotations=torch.nn.functional.normalize(p["rotation"]),
        shs=torch.cat([p["features_dc"], p["features_rest"]], 1),
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
        bg=torch.tensor([0.1, 0.2, 0.3]), W=W, H=H, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2),
        sh_degree=sh_degree)
    return d, cam


def oracle_kwargs(d, **over):
    kw = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    kw.update(over)
    return kw


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
