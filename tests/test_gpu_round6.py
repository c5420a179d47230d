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


def test_chain_rule_with_long_lists_per_scan_block_spatially_coherent_storage_order():
    """The multi-view chain rule cuts every 1024-Gaussian scan block's list of touched Gaussians into groups of 256, one
    workgroup each (round 6).  A model stored in RANDOM order puts ~200 entries into every block (one group); the same Gaussians
    stored in Morton order of their positions put up to 1024 into some blocks and none into most -- all four groups of a block,
    and empty workgroups, then really occur.  Same Gaussians, same six views, same pixel gradients: the parameter gradients and
    the densification statistics must be the permutation of each other."""
    import numpy as np
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.gaussian_model import GaussianModel
    from helpers import rel_l2
    P, W, H = 800_000, 400, 300
    raw = synth.synth_gaussians(P, seed=12, width=W, height=H)
    q = ((raw["xyz"] - raw["xyz"].min(0).values) / (raw["xyz"].max(0).values - raw["xyz"].min(0).values) * 1023).long().clamp(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    perm = torch.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), stable=True)
    pairs = synth.synth_view_set(W, H, device="cuda")
    bg = torch.zeros(3, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=3, device="cuda")

    def run(order):
        t = {n: (v if order is None else v[order]).contiguous() for n, v in raw.items()}
        model = GaussianModel.from_tensors(t["xyz"], t["features_dc"], t["features_rest"], t["scaling"], t["rotation"],
                                           t["opacity"], sh_degree=1, active_sh_degree=1, device="cuda")
        model.init_densification_stats()
        fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False, seg1_fraction=0.0)
        views = [(c, 2 * k + r, r == 0) for k, (cam, scam, _) in enumerate(pairs) for r, c in enumerate((cam, scam))]
        fr.fit_capacity([(c, s) for c, s, _ in views], bg)
        outs = fr.render_batch(views, bg)
        o_t, g_t = [], []
        for o, (_, _, prim) in zip(outs, views):
            o_t.append(o["render"]); g_t.append(gc)
            if prim:
                o_t += [o["rendered_depth"], o["rendered_alpha"]]; g_t += [gd, ga]
        torch.autograd.backward(o_t, g_t)
        torch.cuda.synchronize()
        assert int(fr.overflow_flag.item()) == 0
        return ([p.grad.detach().cpu().numpy() for p in model.parameters()], model.xyz_gradient_accum.cpu().numpy().ravel(),
                model.denom.cpu().numpy().ravel())

    ga_r, acc_r, den_r = run(None)
    ga_m, acc_m, den_m = run(perm)
    pn = perm.numpy()
    assert np.array_equal(den_m, den_r[pn]) and den_r.max() >= 1
    # (the bars of the oracle comparisons: another storage order is another order of the fp32 atomic sums, and equal depth keys --
    # ties keep their index order -- blend in another order)
    assert rel_l2(acc_m, acc_r[pn]) < 2e-4
    touched = np.abs(ga_r[0]).sum(1) > 0
    assert 0.02 < touched.mean() < 0.9
    # the Morton-ordered run really had scan blocks whose lists exceed one group of 256 (and blocks with none)
    per_block = np.add.reduceat((np.abs(ga_m[0]).sum(1) > 0).astype(np.int64), np.arange(0, P, 1024))
    assert per_block.max() > 256 and per_block.min() == 0
    for a, b, n in zip(ga_m, ga_r, "xyz f_dc f_rest scaling rotation opacity".split()):
        assert rel_l2(a, b[pn]) < 2e-4, n
