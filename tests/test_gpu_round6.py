"""GPU, round 6: the compiled `_C` module is what runs behind the reference-shaped surfaces -- the autograd nodes of the module
surface and of the loss functions are C++ nodes (no Python `autograd.Function` on those paths), `_RasterizeGaussians.apply` keeps
the upstream's nine-argument calling convention, and errors raised inside a C++ node on the engine's thread keep their type."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P=3000, W=96, H=64, seed=4):
    from binocular3dgs_amd import synth
    model = synth.synth_model(P, seed=seed, device="cuda", width=W, height=H)
    cam = synth.synth_cameras(W, H, yaws=(2.0,), device="cuda")[0]
    return model, cam, W, H


def test_module_surface_and_loss_functions_run_on_cpp_autograd_nodes():
    from binocular3dgs_amd import _C
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    from binocular3dgs_amd.loss_utils import SmoothLoss, l1_loss, ssim
    from binocular3dgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    assert _C.__file__.endswith(".so")
    model, cam, W, H = _scene()
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device="cuda"), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 1, cam.camera_center, False, False)
    m2d = torch.zeros_like(model.get_xyz, requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(rs)(means3D=model.get_xyz, means2D=m2d, shs=model.get_features,
                                                        opacities=model.get_opacity, scales=model.get_scaling,
                                                        rotations=model.get_rotation)
    # (a C++ node shows in python as `CppFunction`; its name() is the C++ type: torch::autograd::CppNode<b3::RasterizeFn>)
    assert type(color.grad_fn).__name__ == "CppFunction" and "CppNode<b3::RasterizeFn>" in color.grad_fn.name()
    assert radii.dtype == torch.int32 and not radii.requires_grad and depth.grad_fn is color.grad_fn
    gt = torch.rand(3, H, W, device="cuda")
    a, b = l1_loss(color, gt), ssim(color, gt)
    d4 = (depth + 0.5)[None]
    c = SmoothLoss().forward(disparity=d4, image=gt[None])
    w = inverse_warp_images(color[None], 0.3 * d4)
    for t, name in ((a, "L1Fn"), (b, "SsimFn"), (c, "SmoothFn"), (w, "WarpFn")):
        assert type(t.grad_fn).__name__ == "CppFunction" and f"CppNode<b3::{name}>" in t.grad_fn.name(), t.grad_fn.name()
    (a + 0.2 * (1 - b) + 0.05 * c + w.mean() + alpha.mean()).backward()
    assert m2d.grad is not None and float(m2d.grad.abs().max()) > 0 and model._xyz.grad is not None
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_upstream_nine_argument_apply_is_the_exact_forward_and_differentiates():
    from binocular3dgs_amd import rasterizer as R
    model, cam, W, H = _scene(seed=6)
    rs = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device="cuda"), 1.0,
                                         cam.world_view_transform, cam.full_proj_transform, 1, cam.camera_center, False, False)
    e = torch.empty(0, device="cuda")
    m2d = torch.zeros_like(model.get_xyz, requires_grad=True)
    R._lazy.pending.clear()
    out = R._RasterizeGaussians.apply(model.get_xyz, m2d, model.get_features, e, model.get_opacity, model.get_scaling,
                                      model.get_rotation, e, rs)
    assert len(out) == 4 and R._lazy.pending == []          # nine arguments: nothing sync-free, N was read back
    ref = R.GaussianRasterizer(rs)(means3D=model.get_xyz, means2D=m2d, shs=model.get_features, opacities=model.get_opacity,
                                   scales=model.get_scaling, rotations=model.get_rotation)
    for x, y in zip(out, ref):
        assert torch.equal(x, y)
    out[0].sum().backward()          # only the colour differentiated: depth / alpha gradients reach the kernels as NULL
    assert float(m2d.grad.abs().max()) > 0


def test_an_error_inside_a_cpp_node_keeps_its_python_type_through_the_engine():
    """The backward of the module surface with a state buffer that does not belong to it: the library refuses (B3GS_ERR_ARG),
    raised by the C++ node on autograd's device thread -- it must arrive as binocular3dgs_amd._lib.B3gsError with the library's
    message, not as a bare RuntimeError."""
    from binocular3dgs_amd import _C, _lib
    model, cam, W, H = _scene(seed=8)
    e = torch.empty(0, device="cuda")
    bg = torch.zeros(3, device="cuda")
    with pytest.raises(_lib.B3gsError, match="rasterize_gaussians|b3gs_forward|B3GS_ERR"):
        _C.rasterize_gaussians(bg, model.get_xyz, e, model.get_opacity, model.get_scaling, model.get_rotation, 1.0, e,
                               cam.world_view_transform, cam.full_proj_transform, 0.5, 0.5, 0, 0, model.get_features, 1,
                               cam.camera_center, False, False)                              # W = H = 0
    x = torch.rand(1, 3, 8, 8, device="cuda", requires_grad=True)
    from binocular3dgs_amd.loss_utils import ssim
    v = ssim(x, torch.rand(1, 3, 8, 8, device="cuda"))
    with pytest.raises(RuntimeError):                          # a gradient of the wrong shape: refused by the engine itself
        v.backward(torch.ones(2, device="cuda"))
