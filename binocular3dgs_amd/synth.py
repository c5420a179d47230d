"""Deterministic synthetic scenes: the measurement inputs of BASELINE.md section 3 /
SURVEY.md section 8(d).  Everything is generated on the CPU from a seeded torch.Generator so
that every rank, the oracle and the HIP path see identical bits."""
from __future__ import annotations

import math

import torch

from .camera import Camera, look_at_orbit
from .gaussian_model import GaussianModel, inverse_sigmoid

C0 = 0.28209479177387814
YAWS_6 = (0.0, 8.0, -8.0)
SHIFTS_6 = (0.25, -0.15, 0.35)
YAWS_8 = (0.0, 6.0, -6.0, 12.0, -12.0, 18.0, -18.0, 24.0)


def fovy_from(fovx: float, width: int, height: int) -> float:
    return 2.0 * math.atan(math.tan(fovx * 0.5) * height / width)


def synth_gaussians(P: int, seed: int = 0, width: int = 800, height: int = 600, fovx_deg: float = 60.0, K: int = 4,
                    scale_mult: float = 1.0):
    """Raw (pre-activation) parameters, float32 CPU tensors, keyed like the model attributes.  scale_mult multiplies the
    median splat size (1.0 = BASELINE.md section 3; 3.0 = the instance-heavy variant: ~9x the screen area per splat)."""
    g = torch.Generator().manual_seed(seed)
    fovx = math.radians(fovx_deg)
    fovy = fovy_from(fovx, width, height)
    z = 2.0 + 8.0 * torch.rand(P, generator=g)
    x = z * math.tan(fovx / 2) * (2.3 * torch.rand(P, generator=g) - 1.15)
    y = z * math.tan(fovy / 2) * (2.3 * torch.rand(P, generator=g) - 1.15)
    xyz = torch.stack([x, y, z], dim=1)
    scaling = math.log(0.02 * scale_mult) + 0.6 * torch.randn(P, 3, generator=g)   # log of LogNormal(ln .02, .6)
    rotation = torch.randn(P, 4, generator=g)
    opacity = inverse_sigmoid(0.05 + 0.9 * torch.rand(P, 1, generator=g))
    features_dc = torch.randn(P, 1, 3, generator=g) * (0.25 / C0)
    features_rest = torch.randn(P, K - 1, 3, generator=g) * 0.05
    return dict(xyz=xyz.float(), features_dc=features_dc.float(), features_rest=features_rest.float(),
                scaling=scaling.float(), rotation=rotation.float(), opacity=opacity.float())


def synth_model(P: int, seed: int = 0, device="cpu", width=800, height=600, fovx_deg=60.0, K=4, requires_grad=True,
                scale_mult: float = 1.0):
    import math as _m
    sh_degree = int(round(_m.sqrt(K))) - 1
    p = synth_gaussians(P, seed, width, height, fovx_deg, K, scale_mult)
    return GaussianModel.from_tensors(p["xyz"], p["features_dc"], p["features_rest"], p["scaling"], p["rotation"],
                                      p["opacity"], sh_degree=sh_degree, active_sh_degree=min(1, sh_degree),
                                      device=device, requires_grad=requires_grad)


def synth_cameras(width=800, height=600, fovx_deg=60.0, yaws=YAWS_6, device="cpu", yaw_offset=0.0):
    fovx = math.radians(fovx_deg)
    fovy = fovy_from(fovx, width, height)
    cams = []
    for i, yaw in enumerate(yaws):
        R, T = look_at_orbit(yaw + yaw_offset)
        cams.append(Camera(R, T, fovx, fovy, width, height, uid=i, device=device))
    return cams


def synth_view_set(width=800, height=600, fovx_deg=60.0, device="cpu", yaw_offset=0.0):
    """The 6 views of one training iteration: 3 input views + their 3 binocular partners,
    ordered (input0, shifted0, input1, shifted1, input2, shifted2)."""
    views = []
    for cam, t in zip(synth_cameras(width, height, fovx_deg, YAWS_6, device, yaw_offset), SHIFTS_6):
        views.append((cam, cam.shifted(t), t))
    return views


def synth_pixel_grads(width, height, seed=0, device="cpu"):
    """Seeded upstream gradients for backward timing / parity: N(0,1)/(3HW), N(0,1)/HW, N(0,1)/HW."""
    g = torch.Generator().manual_seed(1000 + seed)
    hw = width * height
    gc = torch.randn(3, height, width, generator=g) / (3 * hw)
    gd = torch.randn(1, height, width, generator=g) / hw
    ga = torch.randn(1, height, width, generator=g) / hw
    return gc.to(device), gd.to(device), ga.to(device)
